"""bt2g_results_pack: the packed records the host formats SAM from must carry exactly the bytes of the fixed-stride
records (header, alignment heads, the first nned edits), at the offsets the scan reports."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from bt2test import encode, synth_reads

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("khits", [1, 3])
def test_results_pack_matches_strided_records(khits):
    import torch
    import bowtie2_amd as b
    refs, cur = [], None
    for line in open(os.path.join(GOLD, "tiny.fa")):
        if line.startswith(">"):
            cur = [line[1:].strip(), ""]
            refs.append(cur)
        else:
            cur[1] += line.strip()
    reads = synth_reads([tuple(r) for r in refs], 700, 100, seed=11, sub=0.03, ins=0.004, dele=0.004, len_jitter=30)
    reads += [("junk%d" % i, "ACGT" * 20, "I" * 80) for i in range(5)]          # unaligned reads: header-only records
    ctx = b.Context(0)
    info = ctx.load_index(os.path.join(GOLD, "tiny_s"))
    batch = ctx.upload_reads([encode(s) for _, s, _ in reads], [q.encode() for _, _, q in reads])
    n = len(reads)
    P = b.AlignParams(mm_type=3, mm_max=6, mm_min=2, n_pen=1, rdgapo=5, rdgape=3, rfgapo=5, rfgape=3, gapbar=4, match_bonus=0,
                      khits=khits, mhits=0 if khits > 1 else 50, max_dp_streak=15, max_ug=300, max_dp=300, max_iters=400,
                      n_seed_rounds=2, seed_boost_thresh=300, tighten=3, maxhalf=15, nofw=0, norc=0, do_exact_upfront=1,
                      do_1mm_upfront=1, do_ungapped=1, do_extend=1, large_index=1 if info.off_size == 8 else 0)
    rp = np.zeros(n, dtype=[("minsc", "<i4"), ("interval", "<i4"), ("nceil", "<i4"), ("seedlen", "<i4"), ("seed", "<u4"), ("filt", "<u4")])
    for i, (_, s, _) in enumerate(reads):
        L = len(s)
        rp[i] = (int(-0.6 + -0.6 * L), max(1, int(1 + 1.15 * math.sqrt(L))), int(0.15 * L), 22, 12345 + i, 15)
    rp_t = torch.from_numpy(rp.view(np.uint8).copy()).cuda()
    res, stride = ctx.align_batch(batch, rp_t, P, max(len(s) for _, s, _ in reads))
    packed, offs = ctx.results_pack(res, n, khits)
    torch.cuda.synchronize()
    res = res.cpu().numpy().reshape(n, stride)
    packed = packed.cpu().numpy()
    offs = offs.cpu().numpy()
    head = C.sizeof(b.ReadResult) - C.sizeof(b.Aln)
    aln_sz = C.sizeof(b.Aln)
    aln_head = b.Aln.ned.offset
    nned_off = b.Aln.nned.offset
    assert offs[0] == 0 and np.all(np.diff(offs) >= head)
    n_aln = n_edit = 0
    for i in range(n):
        rec = res[i]
        rr = b.ReadResult.from_buffer_copy(rec[:C.sizeof(b.ReadResult)].tobytes())
        p = int(offs[i])
        assert packed[p:p + head].tobytes() == rec[:head].tobytes(), i
        p += head
        na = rr.nreport if rr.aligned else 0
        assert na <= khits
        for k in range(na):
            a = rec[head + k * aln_sz: head + (k + 1) * aln_sz]
            nned = int(a[nned_off]) | (int(a[nned_off + 1]) << 8)
            keep = aln_head + 6 * nned
            assert packed[p:p + keep].tobytes() == a[:keep].tobytes(), (i, k)
            p += (keep + 7) & ~7
            n_aln += 1
            n_edit += nned
        assert p == int(offs[i + 1]), i
    assert n_aln >= 600 and n_edit > 1000      # the case exercised alignments with edits
    assert int(offs[n]) < n * stride // 4      # and packing actually shrinks the hand-over


@pytest.mark.gpu
def test_align_batch_seed_bound_from_caller():
    """bt2g_align_params::max_seeds: with the caller's bound bt2g_align_batch does not wait for the device to count the seed
    positions.  The result records must not depend on it -- not even on a bound that is too small (reads with more seed
    positions than the tables hold search their seeds inline)."""
    import torch
    import bowtie2_amd as b
    refs, cur = [], None
    for line in open(os.path.join(GOLD, "tiny.fa")):
        if line.startswith(">"):
            cur = [line[1:].strip(), ""]
            refs.append(cur)
        else:
            cur[1] += line.strip()
    reads = synth_reads([tuple(r) for r in refs], 400, 100, seed=17, sub=0.03, ins=0.004, dele=0.004, len_jitter=40)
    ctx = b.Context(0)
    info = ctx.load_index(os.path.join(GOLD, "tiny_s"))
    batch = ctx.upload_reads([encode(s) for _, s, _ in reads], [q.encode() for _, _, q in reads])
    n = len(reads)
    rp = np.zeros(n, dtype=[("minsc", "<i4"), ("interval", "<i4"), ("nceil", "<i4"), ("seedlen", "<i4"), ("seed", "<u4"), ("filt", "<u4")])
    for i, (_, s, _) in enumerate(reads):
        L = len(s)
        rp[i] = (int(-0.6 + -0.6 * L), max(1, int(1 + 1.15 * math.sqrt(L))), int(0.15 * L), 22, 777 + i, 15)
    rp_t = torch.from_numpy(rp.view(np.uint8).copy()).cuda()
    true_bound = max(1 + max(0, len(s) - 22) // int(rp[i]["interval"]) for i, (_, s, _) in enumerate(reads))
    outs = []
    for bound in (0, true_bound, 3, 64):
        P = b.AlignParams(mm_type=3, mm_max=6, mm_min=2, n_pen=1, rdgapo=5, rdgape=3, rfgapo=5, rfgape=3, gapbar=4, match_bonus=0,
                          khits=1, mhits=50, max_dp_streak=15, max_ug=300, max_dp=300, max_iters=400,
                          n_seed_rounds=2, seed_boost_thresh=300, tighten=3, maxhalf=15, nofw=0, norc=0, do_exact_upfront=1,
                          do_1mm_upfront=1, do_ungapped=1, do_extend=1, large_index=1 if info.off_size == 8 else 0, max_seeds=bound)
        res, stride = ctx.align_batch(batch, rp_t, P, max(len(s) for _, s, _ in reads))
        torch.cuda.synchronize()
        rec = res.cpu().numpy().reshape(n, stride)
        # alignment part of the records (status/flags, scores, the reported alignments); the work counters legitimately differ
        # between pre-computed and inline seed search only in where the BW operations were counted
        keep = np.concatenate([rec[:, :24], rec[:, C.sizeof(b.ReadResult) - C.sizeof(b.Aln):C.sizeof(b.ReadResult) - C.sizeof(b.Aln) + b.Aln.ned.offset]], axis=1)
        outs.append(keep.tobytes())
        assert int(rec[:, 0].max()) == 0      # no read flagged
    assert outs[0] == outs[1] == outs[2] == outs[3]
