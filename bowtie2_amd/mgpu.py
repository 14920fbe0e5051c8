"""bowtie2_amd.mgpu -- one node, N GPUs: the drop-in aligner as one process per GPU.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         -m bowtie2_amd.mgpu [--engine EXE] [--backend nccl|gloo] -- <bowtie2-align options> -x IDX -U reads.fq -S out.sam

Every rank runs the SAME product executable the single-GPU user runs (bowtie2_amd/bin/bowtie2-align-{s,l}) on its own GPU; the
index is replicated, no collective touches the data path.  What the reference does with its per-thread AlnSink output -- one
ordered stream and one summed ReportingMetrics (bt2_search.cpp:4812-4900, outq.h:38, aln_sink.cpp:33-101) -- is done here per rank.

How the input is shared out:

* **by byte range** (plain 4-line FASTQ files given with -U or -1/-2; the N-GPU production case).  Rank r takes the slice
  [size*r/N, size*(r+1)/N) of every reads file and counts its newlines (memchr speed); the counts are all-gathered, which tells
  every rank the number of the first record that starts in its slice and -- for -2 -- where that same record number starts in the mate
  file.  The rank then runs the executable with `--shard-bytes a:b[,a2:b2]`: it parses its own bytes and nobody else's, aligns
  them, and writes its SAM piece to a file.  Contiguous read blocks per rank = the reference's -s/-u windows.  The merged output is
  the concatenation of the pieces in rank order: rank 0's executable writes straight into the output file, and rank 0 appends the
  other pieces with sendfile() as their ranks finish.  No SAM text passes through Python memory or the process group.
* **by blocks** (fallback: gzip'ed or piped input, other formats, -s/-u, comma-separated lists, mixed -U with -1/-2): `--shard r/N`
  -- every rank parses the whole input and keeps blocks r, r+N, ... of --batch reads; the SAM text of the blocks is gathered to
  rank 0 over the process group (RCCL on GPUs, gloo in the CPU tests) and written in input order.

Either way the summary counters are all-reduced and rank 0 prints the reference's alignment summary; the result is byte-identical
to the 1-rank run (tests/test_multi_gpu_cpu.py).
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

from . import shard

CHUNK = 1 << 28          # bytes of SAM text per rank per gather round (block mode)
SCAN = 1 << 24           # bytes per read() while counting newlines


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
    blocks, S, P, flagged, parsed = [], [0] * 4, [0] * 10, 0, 0
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
        elif t[0] == "R":
            parsed = int(t[1])
    return blocks, S, P, flagged, parsed


# ---- the user's command line -------------------------------------------------------------------------------------------
VALUE_OPTS = {"-x": "index", "--index": "index", "-U": "unpaired", "--unpaired": "unpaired", "-1": "m1", "-2": "m2", "-S": "out", "--output": "out"}
# options that make the input something other than plain 4-line FASTQ files read from their first record to their last
NOT_BYTE_SHARDABLE = {"-f", "-r", "-c", "-F", "-b", "--tab5", "--tab6", "--12", "--qseq", "--interleaved", "-s", "--skip", "-u", "--upto", "--qupto",
                      "--align-paired-reads", "--preserve-tags"}


def split_args(argv):
    """The user's options without the output option; (args, found) with found = {index, unpaired, m1, m2, out}."""
    args, found = [], {}
    i = 0
    while i < len(argv):
        a = argv[i]
        key, val = a, None
        if a.startswith("--") and "=" in a:
            key, val = a.split("=", 1)
        elif a.startswith("-S=") or a.startswith("-x="):
            key, val = a[:2], a[3:]
        if key in VALUE_OPTS:
            if val is None and i + 1 < len(argv):
                val = argv[i + 1]
                i += 1
            found[VALUE_OPTS[key]] = val
            if VALUE_OPTS[key] != "out":
                args += [key, val]
        else:
            args.append(a)
        i += 1
    return args, found


def usable_cores():
    """Cores this process may run on: the affinity mask, clipped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def default_engine(found):
    here = os.path.dirname(os.path.abspath(__file__))
    base = found.get("index")
    large = base is not None and not os.path.exists(base + ".1.bt2") and os.path.exists(base + ".1.bt2l")
    return os.path.join(here, "bin", "bowtie2-align-l" if large else "bowtie2-align-s")


def plain_file(path):
    if not path or "," in path or path == "-" or not os.path.isfile(path):
        return False
    with open(path, "rb") as f:
        return f.read(2) != b"\x1f\x8b"      # not gzip


def byte_shardable(args, found):
    if any(a.split("=", 1)[0] in NOT_BYTE_SHARDABLE for a in args):
        return None
    if found.get("unpaired") and not (found.get("m1") or found.get("m2")):
        files = [found["unpaired"]]
    elif found.get("m1") and found.get("m2") and not found.get("unpaired"):
        files = [found["m1"], found["m2"]]
    else:
        return None
    return files if all(plain_file(p) for p in files) else None


# ---- byte-range planning -----------------------------------------------------------------------------------------------
def count_newlines(path, a, b):
    n = 0
    with open(path, "rb", buffering=0) as f:
        f.seek(a)
        left = b - a
        while left > 0:
            chunk = f.read(min(SCAN, left))
            if not chunk:
                break
            n += chunk.count(b"\n")
            left -= len(chunk)
    return n


def offset_after_newlines(path, start, k, size):
    """Offset of the byte after the k-th newline at or after `start` (k >= 1); `size` if there are fewer."""
    with open(path, "rb", buffering=0) as f:
        f.seek(start)
        pos = start
        while k > 0:
            chunk = f.read(SCAN)
            if not chunk:
                return size
            c = chunk.count(b"\n")
            if c < k:
                k -= c
                pos += len(chunk)
                continue
            i = -1
            for _ in range(k):
                i = chunk.index(b"\n", i + 1)
            return pos + i + 1
    return start


def plan_byte_ranges(dist, files, rank, world, device):
    """Record-aligned byte range of every reads file for this rank + the number of its first record; None if the files are not
    strict 4-line FASTQ of equal record counts (the caller falls back to block mode; every rank reaches the same verdict)."""
    import torch
    sizes = [os.path.getsize(p) for p in files]
    bounds = [[sz * r // world for r in range(world + 1)] for sz in sizes]
    mine = [count_newlines(p, bounds[k][rank], bounds[k][rank + 1]) for k, p in enumerate(files)]
    # does the file end without a newline?  (then its last line still counts)
    tails = []
    for p, sz in zip(files, sizes):
        t = 0
        if sz > 0 and rank == world - 1:
            with open(p, "rb") as f:
                f.seek(sz - 1)
                t = 0 if f.read(1) == b"\n" else 1
        tails.append(t)
    if dist is not None:
        t = torch.tensor(mine + tails, dtype=torch.int64, device=device)
        allc = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allc, t)
        allc = [[int(v) for v in x.tolist()] for x in allc]
    else:
        allc = [mine + tails]
    nf = len(files)
    cum = [[0] * (world + 1) for _ in range(nf)]         # newlines before the slice of rank q
    for k in range(nf):
        for q in range(world):
            cum[k][q + 1] = cum[k][q] + allc[q][k]
    lines = [cum[k][world] + allc[world - 1][nf + k] for k in range(nf)]
    if any(n % 4 for n in lines) or len(set(lines)) != 1:
        return None
    nrec = lines[0] // 4

    def first_record(r):        # number of the first record that starts at or after rank r's slice of file 0
        if r == 0:
            return 0
        if r >= world:
            return nrec
        return min(nrec, cum[0][r] // 4 + 1)

    def start_of(k, K):         # byte offset of record K in file k
        if K == 0:
            return 0
        if K >= nrec:
            return sizes[k]
        target = 4 * K            # the record starts after this many newlines
        q = 0
        while q + 1 < world and cum[k][q + 1] < target:
            q += 1
        return offset_after_newlines(files[k], bounds[k][q], target - cum[k][q], sizes[k])

    k0, k1 = first_record(rank), first_record(rank + 1)
    return [(start_of(k, k0), start_of(k, k1)) for k in range(nf)], k0, k1 - k0


def gather_bytes(dist, data, device, rank, world):
    """Variable-length byte strings -> list on rank 0 (None elsewhere).  Padded gathers of at most CHUNK bytes per round:
    on GPUs the tensors live in HBM and travel over xGMI (RCCL); with gloo they are host tensors.  (Block mode only.)"""
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


def append_file(dst, path):
    """Append `path` to the open binary file `dst` without passing the bytes through Python objects where the OS allows."""
    with open(path, "rb") as src:
        try:
            dst.flush()
            size = os.fstat(src.fileno()).st_size
            off = 0
            while off < size:
                off += os.sendfile(dst.fileno(), src.fileno(), off, min(1 << 30, size - off))
        except (OSError, AttributeError, ValueError):
            src.seek(off)      # a partial sendfile has already delivered bytes [0, off)
            shutil.copyfileobj(src, dst, 1 << 24)


def run_engine(cmd):
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        return p.returncode, p.stdout, p.stderr.decode(errors="replace")
    except OSError as e:      # e.g. the executable is missing on this rank: still take part in the collectives below
        return 127, b"", "Error: cannot run %s: %s\n" % (cmd[0], e)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    engine, backend, mode = None, None, None
    while argv and argv[0] != "--":
        if argv[0] == "--engine":
            engine = argv[1]; argv = argv[2:]
        elif argv[0] == "--backend":
            backend = argv[1]; argv = argv[2:]
        elif argv[0] == "--sharding":        # bytes | blocks (default: bytes when the input allows it)
            mode = argv[1]; argv = argv[2:]
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
    args, found = split_args(argv)
    out_path = found.get("out")
    if engine is None:
        engine = default_engine(found)
    discord, mixed = "--no-discordant" not in args, "--no-mixed" not in args
    quiet = "--quiet" in args
    # the @PG line of the merged output shows what the user typed, not this rank's rewritten command line
    pg = " ".join([os.path.basename(engine)] + argv)
    files = byte_shardable(args, found) if mode != "blocks" else None
    plan = plan_byte_ranges(dist, files, rank, world, device) if files else None
    # a directory every rank of the node can see; rank 0 names it
    tmp = tempfile.mkdtemp(prefix="bt2g_mgpu_") if rank == 0 else None
    if dist is not None:
        box = [tmp]
        dist.broadcast_object_list(box, src=0)
        tmp = box[0]
    piece, idx, done = (os.path.join(tmp, "piece%d.%s" % (rank, s)) for s in ("sam", "idx", "done"))
    rc = 1
    merge_failed = 0
    try:
        common = [engine] + args + ["--shard-index", idx, "--pg-cmdline", pg]
        if not any(a in ("-p", "--threads") or a.startswith("--threads=") or re.match(r"^-p\d+$", a) for a in args):
            # host threads (FASTQ parsing, SAM formatting) of this rank's executable: its share of the cores the job may use, so that
            # N ranks do not each start as many threads as the node has cores
            # the ranks of THIS node share its cores (a multi-node job has more ranks than that)
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0")) or world
            common += ["-p", str(max(1, usable_cores() // max(1, local_world)))]
        if backend == "nccl":
            common += ["--gpu", str(local_rank)]
        if rank != 0 and "--no-hd" not in args:
            common.append("--no-hd")
        if plan is not None:
            # ---- byte ranges: the pieces concatenate; rank 0's executable writes the output file itself ----
            ranges, first, _ = plan
            direct = rank == 0 and out_path is not None
            target = out_path if direct else piece
            cmd = common + ["--shard-bytes", ",".join("%d:%d" % ab for ab in ranges), "--shard-first-read", str(first), "-S", target]
            ok, S, P, flagged, parsed = False, [0] * 4, [0] * 10, 0, 0
            try:
                code, _, err = run_engine(cmd)
                sys.stderr.write(err)
                if os.path.exists(idx):
                    _, S, P, flagged, parsed = parse_index(idx)
                    ok = code == 0 or (code == 1 and flagged > 0)
            finally:
                # the done-file appears complete or not at all (rank 0 polls for it), whatever happened above
                with open(done + ".tmp", "w") as df:
                    df.write("1" if ok else "0")
                os.replace(done + ".tmp", done)
            all_ok = ok
            if rank == 0:
                f = None
                try:
                    f = open(out_path, "ab") if out_path else sys.stdout.buffer
                    if not direct and ok:
                        append_file(f, piece)
                    deadline = time.time() + float(os.environ.get("BT2G_MGPU_WAIT_S", "14400"))
                    for r in range(1, world):
                        other = os.path.join(tmp, "piece%d." % r)
                        state = ""
                        while state not in ("0", "1"):      # rank r is still aligning; its piece is appended as soon as it ends
                            if os.path.exists(other + "done"):
                                state = open(other + "done").read()
                            if state not in ("0", "1"):
                                if time.time() > deadline:
                                    sys.stderr.write("Error: rank %d did not finish its shard within BT2G_MGPU_WAIT_S\n" % r)
                                    state = "0"
                                else:
                                    time.sleep(0.05)
                        if state != "1":
                            all_ok = False
                        if all_ok:
                            append_file(f, other + "sam")
                except OSError as e:
                    sys.stderr.write("Error: merging the shards failed: %s\n" % e)
                    all_ok = False
                finally:
                    if out_path and f is not None:
                        f.close()
            merge_failed = shard.reduce_sum(dist, [0 if all_ok else 1], device)[0]      # a failed merge on rank 0 fails every rank
            counters = shard.reduce_sum(dist, S + P + [flagged, 0 if ok else 1, parsed], device)
            if os.environ.get("BT2G_MGPU_REPORT_BYTES") and rank == 0:       # test aid: how many bytes of reads files each rank took in
                sys.stderr.write("[mgpu] sharding=bytes parsed_bytes_this_rank=%d parsed_bytes_all=%d input_bytes=%d\n"
                                 % (parsed, counters[16], sum(os.path.getsize(p) for p in files)))
        else:
            # ---- blocks: every rank reads the whole input and keeps its blocks; SAM text is gathered over the process group ----
            cmd = common + ["--shard", "%d/%d" % (rank, world), "-S", piece]
            code, _, err = run_engine(cmd)
            sys.stderr.write(err)
            blocks, S, P, flagged, parsed = parse_index(idx) if os.path.exists(idx) else ([], [0] * 4, [0] * 10, 0, 0)
            ok = os.path.exists(idx) and (code == 0 or (code == 1 and flagged > 0))
            data = open(piece, "rb").read() if ok else b""
            hdr_len = len(data) - sum(b for _, b in blocks)
            counters = shard.reduce_sum(dist, S + P + [flagged, 0 if ok else 1, parsed], device)
            if os.environ.get("BT2G_MGPU_REPORT_BYTES") and rank == 0:
                sys.stderr.write("[mgpu] sharding=blocks parsed_bytes_this_rank=%d parsed_bytes_all=%d\n" % (parsed, counters[16]))
            if dist is None:
                tables, pieces = [("%d\n" % hdr_len + "".join("%d %d\n" % b for b in blocks)).encode()], [data]
            else:
                table = ("%d\n" % hdr_len + "".join("%d %d\n" % b for b in blocks)).encode()
                tables = gather_bytes(dist, table, device, rank, world)
                pieces = gather_bytes(dist, data, device, rank, world)
            if rank == 0 and not counters[15]:
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
        rc = 1 if (counters[15] or merge_failed) else 0
        if rank == 0 and rc == 0:
            if not quiet:
                print_summary(counters[0:4], counters[4:14], counters[4] > 0, discord, mixed)
            if counters[14]:
                sys.stderr.write("Error: %d read(s) exceeded a limit of this build; their SAM records may differ from bowtie2's\n" % counters[14])
                rc = 1
    finally:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        for fn in (piece, idx, done):
            try:
                os.remove(fn)
            except OSError:
                pass
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
