"""bowtie2_amd -- host-side Python mirror of the C ABI in include/bt2g.h.

The product is libbt2g.so (HIP/gfx950, built in-tree by bowtie2_amd/csrc/Makefile).  This module
is plumbing only: ctypes bindings plus helpers that keep batches in HBM as torch tensors and hand
raw device pointers across the C ABI.  There is no CPU fallback: importing works anywhere, but
every compute call raises Bt2gError if the shared library or a gfx950 device is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbt2g.so")

BT2G_OK = 0
ERRORS = {-1: "BT2G_ERR_NO_DEVICE", -2: "BT2G_ERR_IO", -3: "BT2G_ERR_FORMAT", -4: "BT2G_ERR_ARG",
          -5: "BT2G_ERR_HIP", -6: "BT2G_ERR_NOMEM", -7: "BT2G_ERR_UNSUPPORTED"}


class Bt2gError(RuntimeError):
    pass


class IndexInfo(C.Structure):
    _fields_ = [("off_size", C.c_int32), ("line_rate", C.c_int32), ("off_rate", C.c_int32), ("ftab_chars", C.c_int32),
                ("len", C.c_uint64), ("n_pat", C.c_uint64), ("n_frag", C.c_uint64),
                ("zoff_fw", C.c_uint64), ("zoff_bw", C.c_uint64), ("ebwt_bytes", C.c_uint64),
                ("offs_len", C.c_uint64), ("hbm_bytes", C.c_uint64), ("side_sz", C.c_uint32)]


class Reads(C.Structure):
    _fields_ = [("d_seq", C.c_void_p), ("d_qual", C.c_void_p), ("d_off", C.c_void_p), ("n_reads", C.c_uint32)]


class SweepOut(C.Structure):
    _fields_ = [("top", C.c_uint64 * 2), ("bot", C.c_uint64 * 2), ("mine", C.c_uint32 * 2),
                ("hit", C.c_uint8 * 2), ("pad", C.c_uint8 * 6)]


class SeedHit(C.Structure):
    _fields_ = [("topf", C.c_uint64), ("botf", C.c_uint64), ("topb", C.c_uint64), ("botb", C.c_uint64)]


class Resolved(C.Structure):
    _fields_ = [("joined_off", C.c_uint64), ("tidx", C.c_uint64), ("toff", C.c_uint64), ("tlen", C.c_uint64),
                ("straddled", C.c_uint32), ("steps", C.c_uint32)]


class Scoring(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("match_bonus", "mm_pen_type", "mm_max", "mm_min", "n_pen",
                                          "rd_gap_const", "rd_gap_linear", "rf_gap_const", "rf_gap_linear", "gapbar")]


class DpProblem(C.Structure):
    _fields_ = [("rd_off", C.c_uint64), ("rf_off", C.c_uint64), ("rows", C.c_uint32), ("cols", C.c_uint32),
                ("minsc", C.c_int32), ("kind", C.c_uint32), ("out_off", C.c_uint64)]


class DpOut(C.Structure):
    _fields_ = [("best", C.c_int64), ("lastsolcol", C.c_uint32), ("sat8", C.c_uint32), ("band_lo", C.c_int32), ("band_w", C.c_uint32),
                ("has_matrix", C.c_uint32), ("pad", C.c_uint32)]


DP_EE_U8, DP_EE_I16, DP_LOCAL, DP_EE_I16_BAND = 0, 1, 2, 3


class Mm1Hit(C.Structure):      # bt2g_mm1_hit
    _fields_ = [("top", C.c_uint64), ("bot", C.c_uint64), ("score", C.c_int32), ("epos", C.c_uint16), ("echr", C.c_uint8), ("eqchr", C.c_uint8)]


class AlignParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mm_type", "mm_max", "mm_min", "n_pen", "rdgapo", "rdgape", "rfgapo", "rfgape", "gapbar", "match_bonus",
        "khits", "mhits", "max_dp_streak", "max_ug", "max_dp", "max_iters", "n_seed_rounds", "seed_boost_thresh",
        "tighten", "maxhalf", "nofw", "norc", "do_exact_upfront", "do_1mm_upfront", "do_ungapped", "do_extend",
        "large_index", "all_hits", "seed_mms", "overhang", "paired", "pe_policy", "pe_maxfrag", "pe_minfrag", "pe_flags",
        "max_mate_streak", "det_seeds", "seed_cache_mb", "profile", "max_seeds", "max_dp_cols")]


class ReadParams(C.Structure):
    _fields_ = [("minsc", C.c_int32), ("interval", C.c_int32), ("nceil", C.c_int32), ("seedlen", C.c_int32),
                ("seed", C.c_uint32), ("filt", C.c_uint32)]


MAX_EDITS = 200


class Edit(C.Structure):
    _fields_ = [("pos", C.c_uint16), ("chr", C.c_uint8), ("qchr", C.c_uint8), ("type", C.c_uint8), ("pad", C.c_uint8)]


class Aln(C.Structure):
    _fields_ = [("refoff", C.c_int64), ("reflen", C.c_int64), ("refid", C.c_int32), ("score", C.c_int32),
                ("ns", C.c_int16), ("gaps", C.c_int16), ("edits", C.c_int16), ("bases_aligned", C.c_int16),
                ("refns", C.c_uint16), ("nned", C.c_uint16), ("rdlen", C.c_uint16), ("rdextent", C.c_uint16),
                ("rfextent", C.c_uint16), ("trim5p", C.c_uint16), ("trim3p", C.c_uint16),
                ("fw", C.c_uint8), ("pad", C.c_uint8 * 5), ("ned", Edit * MAX_EDITS)]


class ReadResult(C.Structure):
    _fields_ = [("status", C.c_uint8), ("aligned", C.c_uint8), ("maxed", C.c_uint8), ("filt", C.c_uint8),
                ("exhausted", C.c_uint8), ("has_secbest", C.c_uint8), ("pair_type", C.c_uint8), ("pair_flags", C.c_uint8),
                ("secbest", C.c_int32), ("best", C.c_int32), ("nalns", C.c_uint32), ("nreport", C.c_uint32),
                ("n_ex_iters", C.c_uint32), ("n_ex_dps", C.c_uint32), ("n_ex_ugs", C.c_uint32),
                ("n_dp_fail_streak_max", C.c_uint32), ("n_bwops_seed", C.c_uint32), ("n_bwops_ext", C.c_uint32),
                ("n_redundants", C.c_uint32), ("n_bt_attempts", C.c_uint32),
                ("n_ext_left", C.c_uint32), ("n_ext_right", C.c_uint32), ("n_resolve_steps", C.c_uint32), ("n_sides", C.c_uint32),
                ("pair_best", C.c_int32), ("pair_secbest", C.c_int32), ("n_mate_dps", C.c_uint32), ("pad2", C.c_uint32),
                ("alns", Aln * 1)]


class Counters(C.Structure):
    _fields_ = [("rank_queries", C.c_uint64), ("sa_lookups", C.c_uint64), ("ftab_lookups", C.c_uint64),
                ("dp_cells", C.c_uint64), ("bwops", C.c_uint64)]


class BuildParams(C.Structure):
    _fields_ = [("large_index", C.c_int32), ("off_rate", C.c_int32), ("ftab_chars", C.c_int32), ("write_ref", C.c_int32), ("device", C.c_int32)]


class BuildStats(C.Structure):
    _fields_ = [("len", C.c_uint64), ("n_pat", C.c_uint64), ("n_frag", C.c_uint64), ("rounds_fw", C.c_uint32), ("rounds_bw", C.c_uint32),
                ("tied_fw", C.c_uint64), ("tied_bw", C.c_uint64), ("t_parse", C.c_double), ("t_fw", C.c_double), ("t_bw", C.c_double),
                ("t_write", C.c_double)]


# every symbol include/bt2g.h declares: (name, restype, argtypes)
_vp = C.c_void_p
ABI = [
    ("bt2g_version", C.c_uint32, []),
    ("bt2g_ctx_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("bt2g_ctx_destroy", None, [_vp]),
    ("bt2g_last_error", C.c_char_p, [_vp]),
    ("bt2g_index_load", C.c_int, [_vp, C.c_char_p]),
    ("bt2g_index_info_get", C.c_int, [_vp, C.POINTER(IndexInfo)]),
    ("bt2g_index_refname", C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64)]),
    ("bt2g_exact_sweep", C.c_int, [_vp, C.POINTER(Reads), C.c_int, C.c_int, C.c_uint32, _vp, _vp]),
    ("bt2g_seed_search_exact", C.c_int, [_vp, C.POINTER(Reads), _vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    ("bt2g_resolve_offsets", C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_int, _vp, _vp]),
    ("bt2g_scoring_default", None, [C.POINTER(Scoring)]),
    ("bt2g_dp_out_bytes", C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("bt2g_dp_fill", C.c_int, [_vp, C.POINTER(Scoring), _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    ("bt2g_counters_read", C.c_int, [_vp, C.POINTER(Counters), C.c_int, _vp]),
    ("bt2g_align_profile_read", C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int, _vp]),
    ("bt2g_align_timing_read", C.c_int, [_vp, C.POINTER(C.c_float)]),
    ("bt2g_align_timing_read_on", C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    ("bt2g_align_result_stride", C.c_uint64, [C.c_uint32]),
    ("bt2g_align_batch", C.c_int, [_vp, C.POINTER(Reads), _vp, C.POINTER(AlignParams), C.c_uint32, _vp, _vp]),
    ("bt2g_results_pack", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    ("bt2g_build_params_default", None, [C.POINTER(BuildParams)]),
    ("bt2g_index_build", C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.POINTER(BuildParams), C.POINTER(BuildStats)]),
    ("bt2g_one_mm_search", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("bt2g_index_rows", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("bt2g_cli_params", C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.c_int, C.POINTER(AlignParams), C.POINTER(ReadParams)]),
    ("bt2g_index_build_mem", C.c_int, [C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(C.c_uint64), C.c_uint32, C.c_char_p,
                                       C.POINTER(BuildParams), C.POINTER(BuildStats)]),
]

_lib = None


def lib():
    """Load libbt2g.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Bt2gError("libbt2g.so not built: run `make -C bowtie2_amd/csrc` (or __graft_entry__.build())")
        # torch ships its own HIP runtime; it must be the one already resident when libbt2g.so is
        # mapped so that both share one runtime (device pointers and streams cross this boundary).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(os.environ.get("BT2G_LIB", LIB_PATH))    # BT2G_LIB: try an alternative build of the same library
        for name, res, args in ABI:
            fn = getattr(L, name)   # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(ctx, rc, what):
    if rc != 0:
        msg = lib().bt2g_last_error(ctx)
        raise Bt2gError("%s failed: %s (%s)" % (what, ERRORS.get(rc, rc), msg.decode() if msg else ""))


def _stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Context:
    """One per device: owns the HBM-resident index (bt2g_ctx)."""

    def __init__(self, device=0):
        self._h = _vp()
        rc = lib().bt2g_ctx_create(device, C.byref(self._h))
        if rc != 0:
            raise Bt2gError("bt2g_ctx_create failed: %s" % ERRORS.get(rc, rc))
        self.device = device
        self.info = None

    def close(self):
        if self._h:
            lib().bt2g_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_index(self, base):
        _check(self._h, lib().bt2g_index_load(self._h, base.encode()), "bt2g_index_load")
        info = IndexInfo()
        _check(self._h, lib().bt2g_index_info_get(self._h, C.byref(info)), "bt2g_index_info_get")
        self.info = info
        return info

    def refname(self, tidx):
        nm = C.c_char_p()
        ln = C.c_uint64()
        _check(self._h, lib().bt2g_index_refname(self._h, tidx, C.byref(nm), C.byref(ln)), "bt2g_index_refname")
        return nm.value.decode(), ln.value

    # ---- batches -------------------------------------------------------------------
    def upload_reads(self, seqs, quals=None):
        """seqs: list of bytes with codes 0..4; quals: list of ASCII bytes.  Returns a ReadBatch in HBM."""
        import numpy as np
        import torch
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        seq = np.frombuffer(b"".join(seqs) + b"\x00", dtype=np.uint8)
        if quals is None:
            quals = [b"I" * len(s) for s in seqs]
        qual = np.frombuffer(b"".join(quals) + b"\x00", dtype=np.uint8)
        dev = torch.device("cuda", self.device)
        return ReadBatch(torch.from_numpy(seq.copy()).to(dev), torch.from_numpy(qual.copy()).to(dev),
                         torch.from_numpy(offs.view(np.int64).copy()).to(dev), len(seqs))

    # ---- stages --------------------------------------------------------------------
    def exact_sweep(self, batch, nofw=False, norc=False, mine_max=2):
        import torch
        out = torch.zeros(batch.n * C.sizeof(SweepOut), dtype=torch.uint8, device=batch.seq.device)
        rd = batch.struct()
        _check(self._h, lib().bt2g_exact_sweep(self._h, C.byref(rd), int(nofw), int(norc), mine_max,
                                                out.data_ptr(), _stream_ptr()), "bt2g_exact_sweep")
        return out

    def seed_search_exact(self, batch, seedlen, interval, offset, max_seeds):
        """seedlen/interval/offset: int32 torch tensors [n] on the device."""
        import torch
        out = torch.zeros(batch.n * 2 * max_seeds * C.sizeof(SeedHit), dtype=torch.uint8, device=batch.seq.device)
        rd = batch.struct()
        _check(self._h, lib().bt2g_seed_search_exact(self._h, C.byref(rd), seedlen.data_ptr(), interval.data_ptr(),
                                                      offset.data_ptr(), max_seeds, out.data_ptr(), _stream_ptr()),
               "bt2g_seed_search_exact")
        return out

    def index_rows(self, first, n):
        """int64 tensor [n, 16]: rows first .. first + n - 1 seen through the device layout (bt2g_index_rows)."""
        import torch
        out = torch.zeros((n, 16), dtype=torch.int64, device="cuda:%d" % self.device)
        _check(self._h, lib().bt2g_index_rows(self._h, first, n, out.data_ptr(), _stream_ptr()), "bt2g_index_rows")
        return out

    def one_mm_search(self, batch, rparams, params, sweep, cap=8):
        """rparams: uint8 device tensor holding ReadParams[n]; sweep: the tensor exact_sweep returned.  Returns (hits, counts): uint8 device
        tensors holding Mm1Hit[n*4*cap] and the n*4 list lengths (255 = more than cap)."""
        import torch
        hits = torch.zeros(batch.n * 4 * cap * C.sizeof(Mm1Hit), dtype=torch.uint8, device=batch.seq.device)
        cnt = torch.zeros(batch.n * 4, dtype=torch.uint8, device=batch.seq.device)
        rd = batch.struct()
        _check(self._h, lib().bt2g_one_mm_search(self._h, C.addressof(rd), rparams.data_ptr(), C.addressof(params), sweep.data_ptr(), cap,
                                                  hits.data_ptr(), cnt.data_ptr(), _stream_ptr()), "bt2g_one_mm_search")
        return hits, cnt

    def resolve_offsets(self, rows, qlen, reject_straddle=False):
        """rows: int64 tensor [n] (SA rows), qlen: int32 tensor [n]."""
        import torch
        n = rows.numel()
        out = torch.zeros(n * C.sizeof(Resolved), dtype=torch.uint8, device=rows.device)
        _check(self._h, lib().bt2g_resolve_offsets(self._h, rows.data_ptr(), qlen.data_ptr(), n, int(reject_straddle),
                                                    out.data_ptr(), _stream_ptr()), "bt2g_resolve_offsets")
        return out

    def dp_fill(self, probs, rd, qu, rf, out, scoring=None):
        """probs: uint8 device tensor holding DpProblem[n]; rd/qu/rf: uint8 device tensors; out: uint8 device tensor of output blocks
        (bt2g_dp_fill: the worker's own fills as a stage)."""
        sc = scoring
        if sc is None:
            sc = Scoring()
            lib().bt2g_scoring_default(C.byref(sc))
        n = probs.numel() // C.sizeof(DpProblem)
        _check(self._h, lib().bt2g_dp_fill(self._h, C.byref(sc), probs.data_ptr(), n, rd.data_ptr(), qu.data_ptr(), rf.data_ptr(),
                                           out.data_ptr(), _stream_ptr()), "bt2g_dp_fill")

    def align_batch(self, batch, rparams, params, max_read_len):
        """rparams: uint8 device tensor holding ReadParams[n]; returns a uint8 device tensor of result records."""
        import torch
        stride = lib().bt2g_align_result_stride(params.khits)
        out = torch.zeros(batch.n * stride, dtype=torch.uint8, device=batch.seq.device)
        rd = batch.struct()
        _check(self._h, lib().bt2g_align_batch(self._h, C.byref(rd), rparams.data_ptr(), C.byref(params), max_read_len,
                                                out.data_ptr(), _stream_ptr()), "bt2g_align_batch")
        return out, stride

    def results_pack(self, results, n, khits):
        """Packed copy of align_batch's records: (uint8 tensor, int64 offsets tensor [n+1]); see bt2g_results_pack."""
        import torch
        packed = torch.zeros(results.numel(), dtype=torch.uint8, device=results.device)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=results.device)
        _check(self._h, lib().bt2g_results_pack(self._h, results.data_ptr(), n, khits, packed.data_ptr(), offs.data_ptr(),
                                                 _stream_ptr()), "bt2g_results_pack")
        return packed, offs

    def align_timing(self, on_current_stream=False):
        """ms per kernel of the last align batch (of the last one issued on torch's current stream with on_current_stream):
        dict(sweep, one_mm, seeds, extend, align)"""
        out = (C.c_float * 5)()
        if on_current_stream:
            _check(self._h, lib().bt2g_align_timing_read_on(self._h, _stream_ptr(), out), "bt2g_align_timing_read_on")
        else:
            _check(self._h, lib().bt2g_align_timing_read(self._h, out), "bt2g_align_timing_read")
        return dict(zip(["k_exact_sweep", "k_one_mm", "k_seed_search_exact", "k_extend_hits", "k_align_reads"], [float(x) for x in out]))

    def align_profile(self, reset=False):
        out = (C.c_uint64 * 32)()
        _check(self._h, lib().bt2g_align_profile_read(self._h, out, int(reset), _stream_ptr()), "bt2g_align_profile_read")
        return list(out)

    def counters(self, reset=False):
        c = Counters()
        _check(self._h, lib().bt2g_counters_read(self._h, C.byref(c), int(reset), _stream_ptr()), "bt2g_counters_read")
        return c


def cli_params(args, read_len, large_index=True, both_mates_pass=False):
    """(AlignParams, ReadParams) as the drop-in binary derives them from its command line (bt2g_cli_params): `args` is a list of
    bowtie2-align options, e.g. ["--very-sensitive", "-X", "500", "-1", "a", "-2", "b"]; ReadParams.seed is left 0."""
    av = (C.c_char_p * len(args))(*[a.encode() for a in args])
    P, rp = AlignParams(), ReadParams()
    if lib().bt2g_cli_params(len(args), av, read_len, int(large_index), int(both_mates_pass), C.byref(P), C.byref(rp)) != 0:
        raise Bt2gError("bt2g_cli_params rejected %r" % (args,))
    return P, rp


def _build_params(large, off_rate, ftab_chars, device):
    bp = BuildParams()
    lib().bt2g_build_params_default(C.byref(bp))
    bp.large_index, bp.off_rate, bp.ftab_chars, bp.device = int(bool(large)), off_rate, ftab_chars, device
    return bp


def build_index(fasta_paths, out_base, large=False, off_rate=4, ftab_chars=10, device=0):
    """bowtie2-build on the GPU (bt2g_index_build): FASTA files -> <out_base>.{1,2,3,4,rev.1,rev.2}.bt2[l]."""
    bp = _build_params(large, off_rate, ftab_chars, device)
    st = BuildStats()
    arr = (C.c_char_p * len(fasta_paths))(*[p.encode() for p in fasta_paths])
    rc = lib().bt2g_index_build(arr, len(fasta_paths), out_base.encode(), C.byref(bp), C.byref(st))
    if rc != 0:
        raise Bt2gError("bt2g_index_build failed: %s" % ERRORS.get(rc, rc))
    return st


def build_index_mem(names, seqs, out_base, large=False, off_rate=4, ftab_chars=10, device=0):
    """bt2g_index_build_mem: seqs = bytes / numpy uint8 arrays of ASCII sequence characters held in host memory."""
    import numpy as np
    bp = _build_params(large, off_rate, ftab_chars, device)
    st = BuildStats()
    keep = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
    ptrs = (_vp * len(keep))(*[k.ctypes.data for k in keep])
    lens = (C.c_uint64 * len(keep))(*[k.size for k in keep])
    nm = (C.c_char_p * len(keep))(*[n.encode() for n in names]) if names is not None else None
    rc = lib().bt2g_index_build_mem(nm, ptrs, lens, len(keep), out_base.encode(), C.byref(bp), C.byref(st))
    if rc != 0:
        raise Bt2gError("bt2g_index_build_mem failed: %s" % ERRORS.get(rc, rc))
    return st


class ReadBatch:
    def __init__(self, seq, qual, off, n):
        self.seq, self.qual, self.off, self.n = seq, qual, off, n

    def struct(self):
        return Reads(self.seq.data_ptr(), self.qual.data_ptr(), self.off.data_ptr(), self.n)


def structs_from_tensor(t, ctype):
    """Copy a uint8 device tensor holding an array of C structs to host and view it as ctypes array."""
    b = t.cpu().numpy().tobytes()
    n = len(b) // C.sizeof(ctype)
    return (ctype * n).from_buffer_copy(b)
