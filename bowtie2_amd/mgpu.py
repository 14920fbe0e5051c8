"""bowtie2_amd.mgpu -- one node, N GPUs: the drop-in aligner as one process per GPU.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         -m bowtie2_amd.mgpu [--engine EXE] [--backend nccl|gloo] -- <bowtie2-align options> -x IDX -U reads.fq -S out.sam

Every rank runs the SAME product executable the single-GPU user runs (bowtie2_amd/bin/bowtie2-align-{s,l}) on its own GPU
with `--shard rank/N`: the input is cut into blocks of --batch reads dealt round-robin, the index is replicated, no
collective touches the data path.  What the reference does with its per-thread AlnSink output -- one ordered stream and one
summed ReportingMetrics (bt2_search.cpp:4812-4900, outq.h:38, aln_sink.cpp:33-101) -- happens here over the process group
(RCCL on GPUs, gloo in the CPU tests): the SAM text of every block is gathered to rank 0, which writes the blocks in input
order, and the summary counters are all-reduced.  The result is byte-identical to the 1-rank run.
"""
import os
import subprocess
import sys
import tempfile

from . import shard

CHUNK = 1 << 28          # bytes of SAM text per rank per gather round


def pct(a, b):
    return "%.2f%%" % (100.0 * a / b if b else 0.0)


def print_summary(S, P, paired, discord, mixed, f=sys.stderr):
    """The alignment summary as AlnSink::printAlSumm writes it (aln_sink.cpp:377-528), from summed counters."""
    if not paired:
        nread, n0, nuni, nrep = S
        if nread == 0:
            f.write("0 reads\n")      # an empty run has no "of these" and no section (aln_sink.cpp:364-370)
        else:
            f.write("%d reads; of these:\n" % nread)
            f.write("  %d (%s) were unpaired; of these:\n" % (nread, pct(nread, nread)))
            f.write("    %d (%s) aligned 0 times\n" % (n0, pct(n0, nread)))
            f.write("    %d (%s) aligned exactly 1 time\n" % (nuni, pct(nuni, nread)))
            f.write("    %d (%s) aligned >1 times\n" % (nrep, pct(nrep, nread)))
        f.write("%s overall alignment rate\n" % pct(nuni + nrep, nread))
        return
    npair, conc0, cu1, cu2, crep, ndisc, u00, u1, u2, urep = P
    f.write("%d reads; of these:\n" % npair if npair > 0 else "0 reads\n")
    if npair > 0:
        f.write("  %d (%s) were paired; of these:\n" % (npair, pct(npair, npair)))
        f.write("    %d (%s) aligned concordantly 0 times\n" % (conc0, pct(conc0, npair)))
        f.write("    %d (%s) aligned concordantly exactly 1 time\n" % (cu1, pct(cu1, npair)))
        f.write("    %d (%s) aligned concordantly >1 times\n" % (cu2 + crep, pct(cu2 + crep, npair)))
        if discord:
            f.write("    ----\n")
            f.write("    %d pairs aligned concordantly 0 times; of these:\n" % conc0)
            f.write("      %d (%s) aligned discordantly 1 time\n" % (ndisc, pct(ndisc, conc0)))
        ncd0 = conc0 - ndisc
        if mixed:
            f.write("    ----\n")
            f.write("    %d pairs aligned 0 times concordantly or discordantly; of these:\n" % ncd0)
            f.write("      %d mates make up the pairs; of these:\n" % (ncd0 * 2))
            f.write("        %d (%s) aligned 0 times\n" % (u00, pct(u00, ncd0 * 2)))
            f.write("        %d (%s) aligned exactly 1 time\n" % (u1, pct(u1, ncd0 * 2)))
            f.write("        %d (%s) aligned >1 times\n" % (u2 + urep, pct(u2 + urep, ncd0 * 2)))
    tot_al = (cu1 + cu2 + crep) * 2 + ndisc * 2 + u1 + u2 + urep
    f.write("%s overall alignment rate\n" % pct(tot_al, npair * 2))


def parse_index(path):
    blocks, S, P, flagged = [], [0] * 4, [0] * 10, 0
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "B":
            blocks.append((int(t[1]), int(t[2])))
        elif t[0] == "S":
            S = [int(x) for x in t[1:5]]
        elif t[0] == "P":
            P = [int(x) for x in t[1:11]]
        elif t[0] == "F":
            flagged = int(t[1])
    return blocks, S, P, flagged


def default_engine(args):
    here = os.path.dirname(os.path.abspath(__file__))
    base = None
    for i, a in enumerate(args):
        if a == "-x" and i + 1 < len(args):
            base = args[i + 1]
    large = base is not None and not os.path.exists(base + ".1.bt2") and os.path.exists(base + ".1.bt2l")
    return os.path.join(here, "bin", "bowtie2-align-l" if large else "bowtie2-align-s")


def gather_bytes(dist, data, device, rank, world):
    """Variable-length byte strings -> list on rank 0 (None elsewhere).  Padded gathers of at most CHUNK bytes per round:
    on GPUs the tensors live in HBM and travel over xGMI (RCCL); with gloo they are host tensors."""
    import numpy as np
    import torch
    n = torch.tensor([len(data)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    out = [bytearray() for _ in range(world)] if rank == 0 else None
    rounds = (max(sizes) + CHUNK - 1) // CHUNK if max(sizes) > 0 else 0
    arr = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    for k in range(rounds):
        width = min(CHUNK, max(sizes) - k * CHUNK)
        mine = torch.zeros(width, dtype=torch.uint8, device=device)
        piece = arr[k * CHUNK:k * CHUNK + width]
        if piece.size:
            mine[:piece.size] = torch.from_numpy(piece.copy()).to(device)
        got = [torch.zeros(width, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, got, dst=0)
        if rank == 0:
            for r in range(world):
                take = max(0, min(width, sizes[r] - k * CHUNK))
                if take:
                    out[r] += got[r][:take].cpu().numpy().tobytes()
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    engine, backend = None, None
    while argv and argv[0] != "--":
        if argv[0] == "--engine":
            engine = argv[1]; argv = argv[2:]
        elif argv[0] == "--backend":
            backend = argv[1]; argv = argv[2:]
        else:
            break
    if argv and argv[0] == "--":
        argv = argv[1:]
    import torch
    rank, local_rank, world = shard.env_rank()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    device = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist = shard.init(backend, device)
    # the user's command line, with the output redirected to a per-rank piece
    args, out_path = [], None
    i = 0
    while i < len(argv):
        if argv[i] == "-S" and i + 1 < len(argv):
            out_path = argv[i + 1]; i += 2
        else:
            args.append(argv[i]); i += 1
    if engine is None:
        engine = default_engine(args)
    discord, mixed = "--no-discordant" not in args, "--no-mixed" not in args
    quiet = "--quiet" in args
    tmp = tempfile.mkdtemp(prefix="bt2g_mgpu_r%d_" % rank)
    piece, idx = os.path.join(tmp, "piece.sam"), os.path.join(tmp, "piece.idx")
    cmd = [engine] + args + ["--shard", "%d/%d" % (rank, world), "--shard-index", idx, "-S", piece]
    if backend == "nccl":
        cmd += ["--gpu", str(local_rank)]
    if rank != 0 and "--no-hd" not in args:
        cmd.append("--no-hd")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    sys.stderr.write(p.stderr.decode(errors="replace"))
    blocks, S, P, flagged = parse_index(idx) if os.path.exists(idx) else ([], [0] * 4, [0] * 10, 0)
    # exit status 1 with flagged reads = the run completed but some reads exceeded a limit of this build (reported below)
    ok = os.path.exists(idx) and (p.returncode == 0 or (p.returncode == 1 and flagged > 0))
    data = open(piece, "rb").read() if ok else b""
    hdr_len = len(data) - sum(b for _, b in blocks)
    if dist is None:        # one rank: nothing to merge
        if out_path:
            open(out_path, "wb").write(data)
        else:
            sys.stdout.buffer.write(data)
        if not quiet:
            print_summary(S, P, P[0] > 0, discord, mixed)
        return 0 if ok and p.returncode == 0 else 1
    # ---- merge over the process group ----
    counters = shard.reduce_sum(dist, S + P + [flagged, 0 if ok else 1], device)
    table = ("%d\n" % hdr_len + "".join("%d %d\n" % b for b in blocks)).encode()
    tables = gather_bytes(dist, table, device, rank, world)
    pieces = gather_bytes(dist, data, device, rank, world)
    rc = 1 if counters[15] else 0
    if rank == 0 and rc == 0:
        f = open(out_path, "wb") if out_path else sys.stdout.buffer
        where = {}
        for r in range(world):
            lines = bytes(tables[r]).decode().split("\n")
            pos = int(lines[0])
            if r == 0:
                f.write(bytes(pieces[0][:pos]))              # the header comes from rank 0
            for ln in lines[1:]:
                if ln:
                    b, nb = ln.split()
                    where[int(b)] = (r, pos, int(nb))
                    pos += int(nb)
        for b in sorted(where):
            r, pos, nb = where[b]
            f.write(bytes(pieces[r][pos:pos + nb]))
        if out_path:
            f.close()
        if not quiet:
            print_summary(counters[0:4], counters[4:14], counters[4] > 0, discord, mixed)
        if counters[14]:
            sys.stderr.write("Error: %d read(s) exceeded a limit of this build; their SAM records may differ from bowtie2's\n" % counters[14])
            rc = 1
    dist.barrier()
    dist.destroy_process_group()
    for fn in (piece, idx):
        try:
            os.remove(fn)
        except OSError:
            pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
