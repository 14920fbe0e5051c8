"""Shared helpers for the test-suite: ctypes bindings to the oracle (oracle/liboracle.so),
to the reference shim (oracle/_ref/libbt2ref_s.so, present only where oracle/_ref was built)
and small deterministic data generators.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import random
import subprocess
import hashlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
CACHE_DIR = os.environ.get("BT2_TEST_CACHE", "/tmp/bt2_amd_test_cache")

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


class Ebwt(C.Structure):
    _fields_ = [
        ("off_size", C.c_int), ("off_mask", u64), ("len", u64),
        ("line_rate", C.c_int32), ("off_rate", C.c_int32), ("ftab_chars", C.c_int32), ("flags", C.c_int32),
        ("side_sz", C.c_uint32), ("side_bwt_sz", C.c_uint32), ("side_bwt_len", C.c_uint32),
        ("num_sides", u64), ("ebwt_tot_len", u64),
        ("ftab_len", u64), ("eftab_len", u64), ("offs_len", u64),
        ("n_pat", u64), ("n_frag", u64),
        ("plen", u64p), ("rstarts", u64p), ("ebwt", u8p), ("zoff", u64), ("fchr", u64 * 5),
        ("ftab", u64p), ("eftab", u64p), ("offs", u64p), ("fw", C.c_int),
        ("refnames", C.POINTER(C.c_char_p)), ("n_refnames", C.c_size_t),
    ]


class Ref(C.Structure):
    _fields_ = [
        ("nrecs", u64), ("rec_off", u64p), ("rec_len", u64p), ("rec_first", u8p),
        ("nrefs", u64), ("ref_rec_offs", u64p), ("ref_offs", u64p), ("ref_lens", u64p),
        ("buf", u8p), ("buf_sz", u64),
    ]


class Index(C.Structure):
    _fields_ = [("fwd", Ebwt), ("bwd", Ebwt), ("ref", Ref), ("has_bwd", C.c_int), ("has_ref", C.c_int)]


class SweepOut(C.Structure):
    _fields_ = [("top", u64 * 2), ("bot", u64 * 2), ("mine", C.c_uint32 * 2), ("hit", C.c_uint8 * 2),
                ("nelt", u64), ("bwops", u64), ("nrank", u64)]


class SeedHit(C.Structure):
    _fields_ = [("topf", u64), ("botf", u64), ("topb", u64), ("botb", u64), ("bwops", C.c_uint32), ("nrank", C.c_uint32)]


class Scoring(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("match_bonus", "mm_pen_type", "mm_max", "mm_min", "n_pen",
                                        "rd_gap_const", "rd_gap_linear", "rf_gap_const", "rf_gap_linear", "gapbar")]


class Mm1Hit(C.Structure):      # bt2o_mm1_hit (oracle/bt2_oracle.h)
    _fields_ = [("top", C.c_uint64), ("bot", C.c_uint64), ("score", C.c_int64), ("off5p", C.c_uint32), ("chr", C.c_uint8), ("qchr", C.c_uint8),
                ("fw", C.c_uint8), ("kind", C.c_uint8), ("ebwtfw", C.c_uint8), ("pad", C.c_uint8 * 3)]


class Rng(C.Structure):
    _fields_ = [("a", C.c_uint32), ("c", C.c_uint32), ("last", C.c_uint32), ("lastOff", C.c_uint32), ("inited", C.c_int)]


_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "port"])


def oracle():
    """liboracle.so (plain-C restatement).  Built on demand (gcc only)."""
    global _oracle
    if _oracle is None:
        p = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(os.path.join(ORACLE_DIR, "bt2_oracle.c")):
            build_oracle()
        L = C.CDLL(p)
        L.bt2o_index_load.argtypes = [C.POINTER(Index), C.c_char_p]
        L.bt2o_index_load.restype = C.c_int
        L.bt2o_rank.argtypes = [C.POINTER(Ebwt), u64, C.c_int]; L.bt2o_rank.restype = u64
        L.bt2o_rank4.argtypes = [C.POINTER(Ebwt), u64, u64p]
        L.bt2o_row_l.argtypes = [C.POINTER(Ebwt), u64]; L.bt2o_row_l.restype = C.c_int
        L.bt2o_map_lf.argtypes = [C.POINTER(Ebwt), u64]; L.bt2o_map_lf.restype = u64
        L.bt2o_map_lf1c.argtypes = [C.POINTER(Ebwt), u64, C.c_int]; L.bt2o_map_lf1c.restype = u64
        L.bt2o_ftab_lohi.argtypes = [C.POINTER(Ebwt), u64, u64p, u64p]
        L.bt2o_get_offset.argtypes = [C.POINTER(Ebwt), u64, u64p]; L.bt2o_get_offset.restype = u64
        L.bt2o_joined_to_text_off.argtypes = [C.POINTER(Ebwt), u64, u64, u64p, u64p, u64p, C.c_int, C.POINTER(C.c_int)]
        L.bt2o_exact_sweep.argtypes = [C.POINTER(Ebwt), C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.POINTER(SweepOut)]
        L.bt2o_seed_search_exact.argtypes = [C.POINTER(Ebwt), C.POINTER(Ebwt), C.c_char_p, C.c_size_t, C.POINTER(SeedHit)]
        L.bt2o_ref_get_base.argtypes = [C.POINTER(Ref), u64, u64]; L.bt2o_ref_get_base.restype = C.c_int
        L.bt2o_ref_get_stretch.argtypes = [C.POINTER(Ref), C.c_char_p, u64, C.c_int64, C.c_size_t]
        L.bt2o_rng_init.argtypes = [C.POINTER(Rng), C.c_uint32]
        for n in ("bt2o_rng_next_u32", "bt2o_rng_next_u2"):
            getattr(L, n).argtypes = [C.POINTER(Rng)]; getattr(L, n).restype = C.c_uint32
        L.bt2o_rng_next_u64.argtypes = [C.POINTER(Rng)]; L.bt2o_rng_next_u64.restype = u64
        L.bt2o_rng_next_bool.argtypes = [C.POINTER(Rng)]; L.bt2o_rng_next_bool.restype = C.c_int
        L.bt2o_rng_next_float.argtypes = [C.POINTER(Rng)]; L.bt2o_rng_next_float.restype = C.c_float
        L.bt2o_gen_rand_seed.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32]
        L.bt2o_gen_rand_seed.restype = C.c_uint32
        L.bt2o_scoring_default.argtypes = [C.POINTER(Scoring)]
        L.bt2o_score.argtypes = [C.POINTER(Scoring), C.c_int, C.c_int, C.c_int]; L.bt2o_score.restype = C.c_int
        L.bt2o_sw_fill_ee_u8.argtypes = [C.POINTER(Scoring), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                         C.c_char_p, C.c_char_p, C.c_char_p]
        L.bt2o_sw_fill_ee_u8.restype = C.c_int
        L.bt2o_one_mm_search.argtypes = [C.POINTER(Ebwt), C.POINTER(Ebwt), C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(Scoring), C.c_int, C.c_int64,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Mm1Hit), C.c_int]
        L.bt2o_one_mm_search.restype = C.c_int
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libbt2ref_s.so"))


_refshim = {}


def refshim(large=False):
    """The reference's own classes behind oracle/ref_shim.cpp (only where oracle/_ref is built)."""
    key = "l" if large else "s"
    if key not in _refshim:
        L = C.CDLL(os.path.join(REF_DIR, "libbt2ref_%s.so" % key))
        L.ref_open.argtypes = [C.c_char_p]; L.ref_open.restype = C.c_void_p
        L.ref_close.argtypes = [C.c_void_p]
        L.ref_len.argtypes = [C.c_void_p]; L.ref_len.restype = u64
        L.ref_zoff.argtypes = [C.c_void_p, C.c_int]; L.ref_zoff.restype = u64
        L.ref_rank4.argtypes = [C.c_void_p, C.c_int, u64, u64p]
        L.ref_rank.argtypes = [C.c_void_p, C.c_int, u64, C.c_int]; L.ref_rank.restype = u64
        L.ref_map_lf1c.argtypes = [C.c_void_p, C.c_int, u64, C.c_int]; L.ref_map_lf1c.restype = u64
        L.ref_row_l.argtypes = [C.c_void_p, C.c_int, u64]; L.ref_row_l.restype = C.c_int
        L.ref_ftab_lohi.argtypes = [C.c_void_p, C.c_int, u64, u64p, u64p]
        L.ref_get_offset.argtypes = [C.c_void_p, u64]; L.ref_get_offset.restype = u64
        L.ref_joined_to_text_off.argtypes = [C.c_void_p, u64, u64, u64p, u64p, u64p, C.c_int, C.POINTER(C.c_int)]
        L.ref_get_base.argtypes = [C.c_void_p, u64, u64]; L.ref_get_base.restype = C.c_int
        L.ref_exact_sweep.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, u64p]
        L.ref_seed_round.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, u64p, C.c_int, u64p]
        L.ref_seed_round.restype = C.c_int
        L.ref_sw_fill_ee_u8.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int64,
                                        C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.ref_sw_fill_ee_u8.restype = C.c_int64
        i32p = C.POINTER(C.c_int32)
        L.ref_sw_fill_kind.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int64, i32p, i32p, i32p, C.POINTER(C.c_int)]
        L.ref_sw_fill_kind.restype = C.c_int64
        L.ref_set_match_bonus.argtypes = [C.c_void_p, C.c_int]
        L.ref_one_mm.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_int, u64p]
        L.ref_one_mm.restype = C.c_int
        L.ref_rng_stream.argtypes = [C.c_uint32, C.c_char_p, C.c_int, C.POINTER(C.c_uint32)]
        L.ref_score.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]; L.ref_score.restype = C.c_int64
        _refshim[key] = L
    return _refshim[key]


# ---------------------------------------------------------------- data ----
DNA = "ACGT"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def encode(s):
    """ASCII ACGTN -> bytes 0..4"""
    return bytes("ACGTN".index(c) if c in "ACGTN" else 4 for c in s.upper())


def synth_genome(n_refs=3, total=60000, seed=7, n_frac=0.002, repeats=True):
    """Deterministic multi-reference genome with a few N stretches and repeats."""
    rnd = random.Random(seed)
    refs = []
    per = total // n_refs
    for r in range(n_refs):
        L = per + rnd.randint(-per // 10, per // 10)
        s = [rnd.choice(DNA) for _ in range(L)]
        if repeats:  # plant a few repeated segments so ranges > 1 occur
            for _ in range(6):
                ln = rnd.randint(60, 400)
                a = rnd.randint(0, L - ln - 1)
                b = rnd.randint(0, L - ln - 1)
                s[b:b + ln] = s[a:a + ln]
        nN = int(L * n_frac)
        k = 0
        while k < nN:  # N stretches
            ln = rnd.randint(1, 12)
            a = rnd.randint(0, L - ln - 1)
            for i in range(a, a + ln):
                s[i] = "N"
            k += ln
        refs.append(("ref%d description ignored" % r, "".join(s)))
    return refs


def write_fasta(path, refs):
    with open(path, "w") as f:
        for name, s in refs:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")


def synth_reads(refs, n, length, seed=1, sub=0.01, ins=0.001, dele=0.001, n_rate=0.0, len_jitter=0):
    """SURVEY.md section 8d read generator: uniform position, 50/50 strand, 1% subs, 0.1% indels,
    qualities i.i.d. from {38,38,38,30,20,12}."""
    rnd = random.Random(seed)
    quals = "GGG?5-"
    out = []
    i = 0
    while len(out) < n:
        ri = rnd.randrange(len(refs))
        g = refs[ri][1]
        L = length + (rnd.randint(-len_jitter, len_jitter) if len_jitter else 0)
        if len(g) < L + 10:
            continue
        pos = rnd.randrange(0, len(g) - L - 5)
        frag = g[pos:pos + L + 5]
        if frag.count("N") > 2:
            continue
        s = []
        j = 0
        while len(s) < L and j < len(frag):
            r = rnd.random()
            if r < dele:
                j += 1
                continue
            if r < dele + ins:
                s.append(rnd.choice(DNA))
                continue
            c = frag[j]
            j += 1
            if c == "N":
                c = rnd.choice(DNA)
            if rnd.random() < sub:
                c = rnd.choice([x for x in DNA if x != c])
            if n_rate and rnd.random() < n_rate:
                c = "N"
            s.append(c)
        s = "".join(s)
        if len(s) < L:
            continue
        if rnd.random() < 0.5:
            s = revcomp(s)
        q = "".join(rnd.choice(quals) for _ in range(len(s)))
        out.append(("r%d_%d_%d" % (i, ri, pos), s, q))
        i += 1
    return out


def write_fastq(path, reads):
    with open(path, "w") as f:
        for name, s, q in reads:
            f.write("@%s\n%s\n+\n%s\n" % (name, s, q))


def ref_bin(name):
    return os.path.join(REF_DIR, name)


def build_index(fasta, base, large=False):
    """Build a .bt2/.bt2l index with the reference's own bowtie2-build from oracle/_ref."""
    exe = ref_bin("bowtie2-build-l" if large else "bowtie2-build-s")
    subprocess.check_call([exe, "-q", fasta, base], stdout=subprocess.DEVNULL)


def cached_synth_index(n_refs=3, total=60000, seed=7, large=False):
    """(index base path, refs) for a deterministic synthetic genome; built once per machine."""
    key = "synth_%d_%d_%d_%s" % (n_refs, total, seed, "l" if large else "s")
    d = os.path.join(CACHE_DIR, key)
    base = os.path.join(d, "idx")
    refs = synth_genome(n_refs, total, seed)
    ext = "bt2l" if large else "bt2"
    if not os.path.exists(base + ".rev.2." + ext):
        os.makedirs(d, exist_ok=True)
        write_fasta(os.path.join(d, "genome.fa"), refs)
        build_index(os.path.join(d, "genome.fa"), base, large)
    return base, refs


def sha(b):
    return hashlib.sha256(b).hexdigest()


# ---- unaligned BAM writer for the -b tests (SAM spec 4.2; BGZF = gzip members with a "BC" extra field, 4.1) ----
def bam_record(name, flag, seq, qual, tags=b""):
    """one alignment record, unmapped layout: refID/pos/next = -1, no CIGAR; `qual` = Phred+33 text; `tags` = BAM-encoded optional fields"""
    import struct
    codes = "=ACMGRSVTWYHKDBN"
    nib = [codes.index(c) for c in seq.upper()]
    if len(nib) % 2:
        nib.append(0)
    packed = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(name) + 1, 0, 4680, 0, flag, len(seq), -1, -1, 0)
    body += name.encode() + b"\0" + packed + bytes(ord(c) - 33 for c in qual) + tags
    return struct.pack("<I", len(body)) + body


def write_bam(path, records, block=600, refs=(("chrT", 1000),)):
    """records: bam_record() blobs.  `block` = uncompressed bytes per BGZF block (small, so that records straddle blocks)"""
    import struct
    import zlib
    text = b"@HD\tVN:1.6\tSO:queryname\n"
    raw = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", len(refs))
    for nm, ln in refs:
        raw += struct.pack("<I", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<I", ln)
    raw += b"".join(records)

    def bgzf(chunk):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        data = c.compress(chunk) + c.flush()
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, len(data) + 25)
        return hdr + data + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    with open(path, "wb") as f:
        for i in range(0, len(raw), block):
            f.write(bgzf(raw[i:i + block]))
        f.write(bgzf(b""))      # the end-of-file marker block


# ---- the host-compiled worker (tests/hostsim/hostsim.cpp): one compilation per test session ----
_hostsim_built = {}
# the CPU twin is compiled with the capacities of the worker's largest classes (many alignments + long reads)
HOSTSIM_CLASS_FLAGS = ["-DBT2G_CLASS_BIG_K", "-DBT2G_CLASS_MAX_LEN=2048", "-DBT2G_CLASS_MAX_OFFS=128", "-DBT2G_CLASS_MAX_EDITS=640", "-DBT2G_CLASS_MAX_CANDS=1048576", "-DBT2G_CLASS_MAX_WALK_EDITS=1344"]


def build_hostsim(exe):
    """Compile tests/hostsim/hostsim.cpp (the worker source under a plain main(), test-only) to `exe`.  Every test module used to compile
    its own copy (12 s each, a dozen modules); the first call of a session compiles, later ones copy the binary."""
    import shutil
    import subprocess
    hs = os.path.join(ROOT, "tests", "hostsim")
    first = _hostsim_built.get("exe")
    if first is None or not os.path.exists(first):
        # (written beside the target and renamed into place: a copy of the binary that is still running somewhere is not disturbed)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-w"] + HOSTSIM_CLASS_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-o", exe + ".new",
                               os.path.join(hs, "hostsim.cpp"), os.path.join(ROOT, "bowtie2_amd", "csrc", "bt2g_index.cpp"), "-lz", "-lpthread"])
        os.replace(exe + ".new", exe)
        _hostsim_built["exe"] = exe
    elif os.path.abspath(first) != os.path.abspath(exe):
        shutil.copy2(first, exe + ".new")
        os.replace(exe + ".new", exe)
    return exe
