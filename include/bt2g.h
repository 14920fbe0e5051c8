/*
 * include/bt2g.h -- C ABI of libbt2g.so, the MI355X (gfx950) implementation of
 * bowtie2's per-read multiseed hot path.
 *
 * The reference (BenLangmead/bowtie2 v2.5.5) has no FFI / plugin seam for this
 * path; its only boundaries are the bowtie2-align-{s,l} argv/SAM contract and
 * `extern "C" int bowtie(int, const char**)` (bt2_search.cpp:5223-5386).  The
 * seam where a GPU goes is the per-thread worker
 * `static void multiseedSearchWorker(void*)` (bt2_search.cpp:3094-4254) with its
 * file-static inputs multiseed_ebwtFw/ebwtBw/refs/sc (bt2_search.cpp:1910-1917).
 * Each entry point below names the reference interface it stands in for; the
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions: plain C, POD structs, caller-owned buffers, no exceptions.  All
 * functions return 0 on success or a negative bt2g_status.  Pointers named d_*
 * are DEVICE pointers (HBM, e.g. from hipMalloc or a torch tensor's data_ptr());
 * h_* are host pointers.  `stream` is a hipStream_t passed as void* (NULL =
 * default stream).  Calls are asynchronous w.r.t. the host unless stated; one
 * ctx per device, external synchronisation per stream.  There is NO CPU
 * fallback: every compute entry point fails with BT2G_ERR_NO_DEVICE if no
 * gfx950 device is usable.
 *
 * Offsets (SA rows, text offsets) are uint64_t at this boundary regardless of
 * index width; the kernels run 32-bit arithmetic internally for .bt2 indexes.
 */
#ifndef BT2G_H_
#define BT2G_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	BT2G_OK = 0,
	BT2G_ERR_NO_DEVICE = -1,   /* no HIP device / wrong arch                */
	BT2G_ERR_IO = -2,          /* index file missing / unreadable           */
	BT2G_ERR_FORMAT = -3,      /* not a bowtie2 index we understand          */
	BT2G_ERR_ARG = -4,         /* bad argument                               */
	BT2G_ERR_HIP = -5,         /* HIP runtime error (see bt2g_last_error)    */
	BT2G_ERR_NOMEM = -6,
	BT2G_ERR_UNSUPPORTED = -7  /* option outside the implemented hot path    */
} bt2g_status;

typedef struct bt2g_ctx bt2g_ctx;

/* ---- lifecycle -------------------------------------------------------- */
/* Create a context bound to HIP device `device`. */
int  bt2g_ctx_create(int device, bt2g_ctx **out);
void bt2g_ctx_destroy(bt2g_ctx *ctx);
/* Human-readable text for the last error on this ctx (never NULL). */
const char *bt2g_last_error(const bt2g_ctx *ctx);
/* Library/ABI version: (major<<16)|minor. */
uint32_t bt2g_version(void);

/* ---- index ------------------------------------------------------------ */
typedef struct {
	int32_t  off_size;        /* 4 = .bt2, 8 = .bt2l                         */
	int32_t  line_rate, off_rate, ftab_chars;
	uint64_t len;             /* joined-text length                          */
	uint64_t n_pat, n_frag;   /* # reference sequences, # N-free fragments   */
	uint64_t zoff_fw, zoff_bw;
	uint64_t ebwt_bytes;      /* per direction                               */
	uint64_t offs_len;        /* # SA samples (forward index)                */
	uint64_t hbm_bytes;       /* total HBM occupied by the index             */
	uint32_t side_sz;         /* bytes per rank query: 64 (.bt2) / 128 (.bt2l) */
} bt2g_index_info;

/*
 * Load <base>.{1,2,rev.1,3,4}.bt2 or .bt2l (auto-detected) and make it resident
 * in HBM.  Replaces Ebwt::Ebwt + Ebwt::loadIntoMemory for both directions
 * (bt2_search.cpp:4986,5155,4832-4854; bt2_io.cpp:39-633) and
 * BitPairReference::BitPairReference (reference.cpp:30-264).  Synchronous.
 * The three large sections (BWT sides, SA sample, 2-bit reference) are streamed
 * from the files to the device through pinned buffers by a few reader threads
 * while the sections already there are transcoded (3.1 Gbp .bt2l: ~0.6 s from
 * the page cache); a host that cannot pin the buffers gets the same load through
 * host memory.  The files must not change while the call runs.
 */
int bt2g_index_load(bt2g_ctx *ctx, const char *base);
int bt2g_index_info_get(const bt2g_ctx *ctx, bt2g_index_info *out);
/* Reference names / lengths (for the @SQ header lines); name pointer valid until ctx destroy. */
int bt2g_index_refname(const bt2g_ctx *ctx, uint64_t tidx, const char **name, uint64_t *len);

/* ---- reads in HBM ----------------------------------------------------- */
/*
 * A read batch resident in HBM.  d_seq: concatenated read characters, one
 * byte per base, codes 0..4 = A,C,G,T,N (Read::patFw, read.h:39).  d_qual:
 * ASCII (Phred+33) qualities, same layout.  d_off[n_reads+1]: start offset of
 * each read in d_seq/d_qual (ragged; empty reads allowed).
 */
typedef struct {
	const uint8_t  *d_seq;
	const uint8_t  *d_qual;
	const uint64_t *d_off;
	uint32_t        n_reads;
} bt2g_reads;

/* ---- stage 1: exact end-to-end sweep ----------------------------------- */
/* Per (read, strand) result of SeedAligner::exactSweep (aligner_seed.cpp:856-970). */
typedef struct {
	uint64_t top[2], bot[2];  /* [0] fw read, [1] rc read; valid iff hit     */
	uint32_t mine[2];         /* lower bound on # edits (capped at mine_max) */
	uint8_t  hit[2];          /* exact end-to-end hit recorded               */
	uint8_t  pad[6];
} bt2g_sweep_out;

/*
 * Replaces SeedAligner::exactSweep for a whole batch: d_out[n_reads].
 * nofw/norc as in the reference; mine_max = 2 at the reference's call site
 * (bt2_search.cpp:3514-3526).
 */
int bt2g_exact_sweep(bt2g_ctx *ctx, const bt2g_reads *reads, int nofw, int norc, uint32_t mine_max,
                     bt2g_sweep_out *d_out, void *stream);

/* ---- stage 2: exact multiseed search ----------------------------------- */
typedef struct {
	uint64_t topf, botf;      /* range in the forward index (botf==topf: no hit or seed contains N) */
	uint64_t topb, botb;      /* range in the mirror index                   */
} bt2g_seed_hit;

/*
 * One seeding round with -N 0 for a whole batch: Seed::mmSeeds(0,L) +
 * SeedAligner::instantiateSeeds + searchAllSeeds (aligner_seed.cpp:498-720,
 * 1638-2037).  For read r: nseeds = 1 + (len-offset-L)/interval when
 * len-offset > L else 1 (:523-526); seeds whose window would run off the read
 * are not produced when offset>0 and L+offset > len (bt2_search.cpp:3927).
 * Per-read parameters come from d_seedlen/d_interval/d_offset (each [n_reads]);
 * seedlen is clamped to the read length as the reference does (:229).
 * Output slot for (read r, strand s (0 fw,1 rc), seed i) is
 * d_out[(r*2+s)*max_seeds + i]; slots with i >= nseeds(r) are zeroed.
 */
int bt2g_seed_search_exact(bt2g_ctx *ctx, const bt2g_reads *reads,
                           const uint32_t *d_seedlen, const uint32_t *d_interval, const uint32_t *d_offset,
                           uint32_t max_seeds, bt2g_seed_hit *d_out, void *stream);

/* ---- the index as the device holds it, row by row (test hook) ------------ */
/*
 * The device does not keep the reference's on-disk layout (sides + sampled suffix array, bt2_idx.h:1060-1100): bt2g_index_load
 * transcodes it into 64-byte rank blocks and a full suffix array.  This reads rows first_row .. first_row + n_rows - 1
 * (0 <= row <= len) back through that layout, 16 values per row:
 *   [0] row  [1] Ebwt::getOffset(row)  [2] LF steps the reference's walk takes  [3..6] rank of A,C,G,T at row, forward index
 *   [7] character of the row (-1: the $ row)  [8] mapLF1(row)  [9..12] ranks in the mirror index
 *   [13],[14] rank of character (row & 3) at row and at min(row + 37, len)  [15] sides the reference reads for that pair.
 */
int bt2g_index_rows(bt2g_ctx *ctx, uint64_t first_row, uint64_t n_rows, uint64_t *d_out, void *stream);

/* ---- stage 3: SA-row -> text offset ------------------------------------ */
typedef struct {
	uint64_t joined_off;      /* Ebwt::getOffset(row) (bt2_idx.cpp:150)      */
	uint64_t tidx, toff, tlen;/* Ebwt::joinedToTextOff (bt2_idx.cpp:54); tidx = UINT64_MAX if rejected */
	uint32_t straddled;
	uint32_t steps;           /* LF steps walked                              */
} bt2g_resolved;

/*
 * Batched offset resolution: replaces GroupWalk2S::advanceElement's result
 * (group_walk.h:1160) == Ebwt::getOffset + joinedToTextOff.  d_rows[n],
 * d_qlen[n] (hit length, for the straddle test), reject_straddle as in the
 * reference call (aligner_sw_driver.cpp:1135).
 */
int bt2g_resolve_offsets(bt2g_ctx *ctx, const uint64_t *d_rows, const uint32_t *d_qlen, uint64_t n,
                         int reject_straddle, bt2g_resolved *d_out, void *stream);

/* ---- stage 4: the DP fills ----------------------------------------------- */
typedef struct {
	int32_t match_bonus;      /* 0 in end-to-end mode, --ma in local mode     */
	int32_t mm_pen_type;      /* 3 = quality-aware (default), 1 = constant    */
	int32_t mm_max, mm_min;   /* --mp 6,2                                     */
	int32_t n_pen;            /* --np 1                                       */
	int32_t rd_gap_const, rd_gap_linear, rf_gap_const, rf_gap_linear; /* --rdg 5,3 --rfg 5,3 */
	int32_t gapbar;           /* --gbar 4                                     */
} bt2g_scoring;

void bt2g_scoring_default(bt2g_scoring *sc);

/* One DP problem: the read in the orientation being aligned + the reference window as masks (1 << base, 16 = N). */
typedef struct {
	uint64_t rd_off;          /* offset into d_rd (codes 0..4) and d_qu (ASCII qualities, Phred+33) */
	uint64_t rf_off;          /* offset into d_rf: cols + 1 masks (the column past the window is looked at by the local candidate test) */
	uint32_t rows, cols;
	int32_t  minsc;           /* minimum score: sets the band of the 8-bit fill, the local fill's bail-out column */
	uint32_t kind;            /* BT2G_DP_EE_U8, BT2G_DP_EE_I16 or BT2G_DP_LOCAL */
	uint64_t out_off;         /* offset (bytes, multiple of 8) of this problem's output block in d_out */
} bt2g_dp_problem;
#define BT2G_DP_EE_U8   0     /* alignNucleotidesEnd2EndSseU8  (aligner_swsse_ee_u8.cpp:775-1146)  */
#define BT2G_DP_EE_I16  1     /* alignNucleotidesEnd2EndSseI16 (aligner_swsse_ee_i16.cpp:780-1170) */
#define BT2G_DP_LOCAL   2     /* alignNucleotidesLocalSseU8 / ...I16 (aligner_swsse_loc_u8.cpp:927, aligner_swsse_loc_i16.cpp:938) */
#define BT2G_DP_EE_I16_BAND 3 /* the 16-bit end-to-end kernel's arithmetic on the band of diagonals an alignment can touch, predecessor bits stored: the form the
                                 worker uses for reads whose minimum score is below -254 wherever the band has at most 2 048 diagonals */

/* what a problem's output block starts with */
typedef struct {
	int64_t  best;            /* best score (last row end to end, any cell local), de-biased; INT64_MIN = the band exceeds this build */
	uint32_t lastsolcol;      /* local: last column whose maximum reaches minsc before the fill's bail-out point (lastsolcol_) */
	uint32_t sat8;            /* local: the reference's 8-bit kernel would have saturated (it then redoes the fill in 16 bits) */
	int32_t  band_lo;         /* BT2G_DP_EE_U8: cell (i, j) of the matrix is byte i * band_w + (j - i + band_lo)              */
	uint32_t band_w;
	uint32_t has_matrix;      /* 0: end-to-end fill that did not reach minsc -- no matrix was stored                          */
	uint32_t pad;
} bt2g_dp_out;

/*
 * The fills of the worker (`SwAligner::align`, aligner_sw.cpp:500-729) as a stage: every problem is filled by the same device
 * functions k_align_reads uses, one wavefront per problem, and what they leave behind is written to the problem's output block:
 *   bt2g_dp_out, then
 *   BT2G_DP_EE_U8 : int16 lastrow[cols rounded up to 4] (scores of the last row, de-biased, -255 where the band does not reach),
 *                   then, if has_matrix, rows * band_w bytes of predecessor bits (1 H-diagonal, 2 H==E, 4 H==F, 8 E opens from
 *                   H-left, 16 E extends E-left, 32 F opens from H-up, 64 F extends F-up: what the reference's backtrace asks of
 *                   H/E/F, aligner_swsse_ee_u8.cpp:1330-1520); only cells inside the band are meaningful;
 *   BT2G_DP_EE_I16: int32 H[rows*cols], E[...], F[...] row-major, the cell values of the reference's 16-bit kernel
 *                   (0x7fff = perfect, -32768 = minus infinity);
 *   BT2G_DP_EE_I16_BAND: as BT2G_DP_EE_U8 (last row clamped at -32768, predecessor bits of the band); best = INT64_MIN where the band
 *                   does not fit and the worker falls back to the BT2G_DP_EE_I16 form;
 *   BT2G_DP_LOCAL : rows * cols bytes of predecessor bits, row-major, the same seven questions with the local kernels' `> floor`
 *                   rule folded in (a neighbour whose score is 0 is no predecessor, aligner_swsse_loc_u8.cpp:1530-1660) -- what
 *                   the worker's local fill stores instead of scores; best / lastsolcol / sat8 in the header carry what the worker
 *                   takes from the scores themselves.
 * bt2g_dp_out_bytes() gives the size of a block.  rows <= BT2G_MAX_READ_LEN, cols < 2176 (problems wider than 1099 columns make the launch hold more per-column state: BT2G_MAX_DP_COLS).
 */
uint64_t bt2g_dp_out_bytes(uint32_t kind, uint32_t rows, uint32_t cols);
int bt2g_dp_fill(bt2g_ctx *ctx, const bt2g_scoring *sc, const bt2g_dp_problem *d_probs, uint32_t n,
                 const uint8_t *d_rd, const uint8_t *d_qu, const uint8_t *d_rf, uint8_t *d_out, void *stream);


/* ---- the fused per-read worker ------------------------------------------ */
/*
 * Replaces `static void multiseedSearchWorker(void*)` (bt2_search.cpp:3094-4254) for a whole
 * batch of unpaired reads, or of pairs (params.paired: mates interleaved, see bt2g_align_params): exact end-to-end sweep, 1-mismatch end-to-end search, -N 0 / -N 1 seed
 * rounds, seed-hit prioritisation, offset resolution, DP framing, end-to-end and local SW fill (8/16-bit semantics),
 * backtrace, redundancy checks, -M/-k reporting state and the final selection -- one
 * wavefront per read, the reference's RNG draw order reproduced, so that the SAM written
 * from these records is byte-identical to the reference's.  Scope (rejected otherwise by the
 * host): reads <= BT2G_MAX_READ_LEN, at most BT2G_MAX_KHITS alignments per read (a batch with bt2g_align_params::khits above 64 runs in the worker's
 * many-alignments class: larger per-wave capacities, result records of khits alignments each -- keep such batches small), opposite-mate windows <= 1100 columns, or up to BT2G_MAX_DP_COLS when bt2g_align_params::max_dp_cols asks for it (a read or pair
 * over a limit comes back with status bit 0 set).
 */
#define BT2G_MAX_READ_LEN 1999   /* the reference changes algorithm at 2 000 bp (checkpointed backtrace, aligner_sw.cpp:514: out of scope); a batch whose longest
                                    read is above 512 bp runs in the worker's long-read class (khits <= 64 there) */
#define BT2G_MAX_EDITS    200    /* edits an alignment slot of bt2g_aln holds */
#define BT2G_MAX_EDITS_LONG 640  /* ... of the slots the long-read class writes (reads of 513 ... 1 999 bp: minsc -1 200 at the default --score-min) */
#define BT2G_MAX_KHITS    1000   /* -k ceiling of this build (the reference has none, aln_sink.cpp:33-326); -a reports up to this many and flags a read that has more */

/* what bt2_search.cpp keeps in file statics (:69-266), for the options that reach the worker */
typedef struct {
	int32_t mm_type, mm_max, mm_min, n_pen, rdgapo, rdgape, rfgapo, rfgape, gapbar, match_bonus;
	int32_t khits, mhits;          /* -k ; -M (mhits > 0 => -M mode)                         */
	int32_t max_dp_streak;         /* -D                                                     */
	int32_t max_ug, max_dp, max_iters;
	int32_t n_seed_rounds;         /* nSeedRounds (-R)                                       */
	int32_t seed_boost_thresh, tighten, maxhalf;
	int32_t nofw, norc;
	int32_t do_exact_upfront, do_1mm_upfront, do_ungapped, do_extend;
	int32_t large_index;           /* RNG draws differ in the 64-bit build (aligner_sw_driver.cpp:103-109) */
	int32_t all_hits;              /* -a: no limit on alignments, effort limits lifted, deterministic seed order (khits is then
	                                  only the capacity of the result record; more alignments than that flag the read) */
	int32_t seed_mms;              /* -N: 0 = exact seeds, 1 = one mismatch per seed (Seed::oneMmSeeds)          */
	int32_t overhang;              /* --overhang (gReportOverhangs): DP windows may run past the reference ends by
	                                  the N ceiling, overhanging read ends come back soft-clipped                    */
	/* paired-end mode (extendSeedsPaired, PairedEndPolicy pe.h:169).  With paired != 0 the batch holds mates
	   interleaved (read 2i = mate 1, read 2i+1 = mate 2 of pair i) and result record 2i / 2i+1 belong together */
	int32_t paired;                /* may differ from one bt2g_align_batch call to the next on the same context (a run with -1/-2 AND -U sends
	                                  its pair batches first, then unpaired ones): the context re-initialises its work arena when the mode changes */
	int32_t pe_policy;             /* PE_POLICY_FF 1, RR 2, FR 3 (default), RF 4 (pe.h:33-36)                   */
	int32_t pe_maxfrag, pe_minfrag;/* -X / -I                                                                    */
	int32_t pe_flags;              /* BT2G_PE_* below                                                            */
	int32_t max_mate_streak;       /* maxMateStreak (10), scaled with -k like max_dp_streak                      */
	int32_t det_seeds;             /* -d: seed-hit ranges in sorted order, rows in index order, no sampling (prioritizeSATupsIdxs) */
	int32_t seed_cache_mb;         /* --seed-cache-sz (default 20): size of the reference's per-read seed-hit cache, whose exhaustion on
	                                  reads with millions of seed-hit rows is part of its output (0 = 20)                       */
	int32_t profile;               /* 1: the worker reads the device clock around every phase (what bt2g_align_profile_read reports);
	                                  0: no clock reads, the time slots of the profile stay 0                                    */
	int32_t max_seeds;             /* > 0: an upper bound the caller vouches for on the seed positions per strand of any read in the batch
	                                  (1 + (len - seedlen) / interval for every read): bt2g_align_batch sizes its seed tables from it and
	                                  returns WITHOUT synchronising the stream (reads that exceed it are searched inline by the worker, still
	                                  exact).  0: the bound is computed on the device and bt2g_align_batch waits for it (one 4-byte D2H)   */
	int32_t max_dp_cols;           /* > 1100: DP windows (opposite-mate windows: about -X + read length + 2 x gaps) of up to this many columns occur in
	                                  the batch -- at most BT2G_MAX_DP_COLS; the launch then holds that much per-column state in LDS and runs 12 instead of
	                                  16 waves per CU.  <= 1100 (0: default): windows of up to 1 100 columns; a read or pair that needs a wider one is flagged  */
} bt2g_align_params;
#define BT2G_MAX_DP_COLS 2176
#define BT2G_PE_DOVETAIL_OK  1     /* --dovetail                       */
#define BT2G_PE_CONTAIN_OK   2     /* cleared by --no-contain          */
#define BT2G_PE_OLAP_OK      4     /* cleared by --no-overlap          */
#define BT2G_PE_EXPAND       8     /* gExpandToFrag (always on)        */
#define BT2G_PE_FLIP_OK     16     /* gFlippedMatesOK (always off)     */
#define BT2G_PE_DISCORD     32     /* cleared by --no-discordant       */
#define BT2G_PE_MIXED       64     /* cleared by --no-mixed            */
#define BT2G_PE_MATE1FW    128     /* --fr/--ff: mate 1 forward        */
#define BT2G_PE_MATE2FW    256     /* --ff/--rf                        */

/* per-read inputs the host derives with the reference's formulas (bt2_search.cpp:3352-3450, pat.cpp:45) */
typedef struct {
	int32_t  minsc, interval, nceil, seedlen;
	uint32_t seed;
	uint32_t filt;                 /* bit0 nfilt, bit1 scfilt, bit2 lenfilt, bit3 qcfilt (1 = passes) */
} bt2g_read_params;

typedef struct {                   /* Edit (edit.h:50): pos is w.r.t. the read's 5' end      */
	uint16_t pos;
	uint8_t  chr, qchr;            /* reference / read character (ASCII) or '-'              */
	uint8_t  type;                 /* 1 read gap, 2 ref gap, 3 mismatch (EDIT_TYPE_*)        */
	uint8_t  pad;
} bt2g_edit;

typedef struct {                   /* AlnRes (aligner_result.h:792), the fields SAM needs    */
	int64_t  refoff, reflen;
	int32_t  refid, score;
	int16_t  ns, gaps, edits, bases_aligned;
	uint16_t refns, nned, rdlen, rdextent, rfextent, trim5p, trim3p;
	uint8_t  fw;
	uint8_t  pad[5];
	bt2g_edit ned[BT2G_MAX_EDITS];
} bt2g_aln;

typedef struct {
	uint8_t  status;               /* 0 ok; bit 0 = a fixed-capacity work buffer overflowed (result not reference-identical) */
	uint8_t  aligned, maxed, filt, exhausted, has_secbest;
	uint8_t  pair_type;            /* paired-end: 0 = this mate reported on its own (or unaligned), 1 = concordant, 2 = discordant */
	uint8_t  pair_flags;           /* bit0 pair over the -M ceiling (pairMax), bit1 a second-best pair score exists     */
	int32_t  secbest, best;        /* XS:i / MAPQ inputs                                      */
	uint32_t nalns, nreport;
	uint32_t n_ex_iters, n_ex_dps, n_ex_ugs, n_dp_fail_streak_max, n_bwops_seed, n_bwops_ext, n_redundants, n_bt_attempts;
	uint32_t n_ext_left, n_ext_right, n_resolve_steps, n_sides;   /* seed-hit extension steps, SA-walk steps, sides read */
	int32_t  pair_best, pair_secbest;  /* paired-end MAPQ inputs: best / second-best concordant (or discordant) pair score   */
	uint32_t n_mate_dps, pad2;         /* opposite-mate DPs run for this anchor mate; pad2: bits 0-15 the capacity site that flagged the read (diagnostics), bits 16-31 alignment slot size / 8 */
	bt2g_aln alns[1];              /* nreport (<= khits) entries                              */
} bt2g_read_result;

/* Bytes between consecutive result records for a given -k.  A record is the header of bt2g_read_result followed by `nreport` alignment slots.
 * The slots of one record all have the same size, written by the worker into the record itself: (pad2 >> 16) * 8 bytes -- sizeof(bt2g_aln)
 * except in records of the long-read class, whose slots hold BT2G_MAX_EDITS_LONG edits.  The stride leaves room for either (khits <= 64; the
 * many-alignments class above that has bt2g_aln slots only).  bt2g_results_pack cuts every slot after its last edit: consumers of packed
 * records walk the alignments by their `nned` and never need the slot size. */
uint64_t bt2g_align_result_stride(uint32_t khits);
/*
 * d_rparams[n_reads]; d_results: n_reads records of bt2g_align_result_stride(khits) bytes.
 * max_read_len sizes the per-wave DP scratch (longer reads come back with status 1).
 * Concurrency: a context keeps one working set (work arena, pre-computation tables, queue head) per stream it is called on, up to 4
 * streams; calls on DIFFERENT streams may be made from different threads and their kernels and copies overlap (the reference overlaps
 * I/O and alignment with its read-ahead thread and -p worker threads, pat.h:1287-1301, bt2_search.cpp:4812-4900).  Calls on the same
 * stream are ordered by the stream.
 */
int bt2g_align_batch(bt2g_ctx *ctx, const bt2g_reads *reads, const bt2g_read_params *d_rparams,
                     const bt2g_align_params *params, uint32_t max_read_len,
                     void *d_results, void *stream);

/* ---- stage 2b: 1-mismatch end-to-end search ------------------------------ */
/* One hit of SeedAligner::oneMmSearch (aligner_seed.cpp:975-1325; EEHit, aligner_seed.h:482-560). */
typedef struct {
	uint64_t top, bot;        /* range in the forward index                                   */
	int32_t  score;
	uint16_t epos;            /* mismatch offset from the read's 5' end (Edit::pos)           */
	uint8_t  echr, eqchr;     /* reference character 0..3 / read character 0..4               */
} bt2g_mm1_hit;

/*
 * Replaces SeedAligner::oneMmSearch(..., repex = false, rep1mm = true, ...) as the worker calls it
 * (bt2_search.cpp:3704-3728) for a whole batch.  Input besides the reads: the per-read parameters
 * (minsc, nceil, filters) and the output of bt2g_exact_sweep (the reference only searches a strand
 * whose sweep left mine <= 1).  The search of read r is four independent walks -- strand (0 fw, 1 rc)
 * x index (0 forward, 1 mirror) -- and list l = r*4 + strand*2 + index holds the hits of one walk in
 * the order the reference adds them: d_hits[l*cap .. l*cap + d_n[l]); the reference's hit list of
 * the read is lists 4r .. 4r+3 concatenated.  d_n[l] = 255: the walk found more than `cap` hits
 * (the worker then redoes that read with its inline search).  cap <= 254.  Synchronises `stream`
 * (it frees its scratch).
 */
int bt2g_one_mm_search(bt2g_ctx *ctx, const bt2g_reads *reads, const bt2g_read_params *d_rparams,
                       const bt2g_align_params *params, const bt2g_sweep_out *d_sweep, uint32_t cap,
                       bt2g_mm1_hit *d_hits, uint8_t *d_n, void *stream);

/*
 * Cuts the fixed-stride records of a finished batch down to what the host needs before they cross PCIe
 * (a record is sized for BT2G_MAX_EDITS edits per alignment; a typical read carries a handful).  Packed record
 * i starts at d_packed + d_offsets[i] (d_offsets has n_reads + 1 entries; the last one is the total size) and
 * is the bt2g_read_result header followed by its nreport alignments, each cut after ned[nned - 1] and padded
 * to 8 bytes -- so alns[0] is still addressable through the struct, later ones by walking.  d_packed needs
 * room for n_reads * bt2g_align_result_stride(khits) bytes in the worst case.  Runs on `stream` after
 * bt2g_align_batch.  The reference formats AlnRes objects in place (AlnSinkWrap::finishRead,
 * aln_sink.cpp:620); this is the device-to-host leg of the same hand-over.
 */
int bt2g_results_pack(bt2g_ctx *ctx, const void *d_results, uint32_t n_reads, uint32_t khits,
                      void *d_packed, uint64_t *d_offsets, void *stream);

/*
 * Device time (ms, HIP events on `stream`) of each kernel of the most recent bt2g_align_batch:
 * [0] k_exact_sweep [1] k_one_mm [2] k_seed_search_exact [3] k_extend_hits [4] k_align_reads.
 * Blocks until that batch has finished.  Measurement aid (bench.py's roofline), no reference counterpart.
 */
int bt2g_align_timing_read(bt2g_ctx *ctx, float *out_ms5);
/* the same for the most recent batch issued on `stream` (a context keeps one working set per stream, see bt2g_align_batch) */
int bt2g_align_timing_read_on(bt2g_ctx *ctx, void *stream, float *out_ms5);

/*
 * Device-clock ticks (100 MHz wall clock) the fused worker spent per phase, summed over reads since the last
 * reset: [0] exact sweep [1] 1-mm search [2] seed search [3] rank+prioritise [4] offset resolution
 * [5] ref fetch + DP fill [6] gather + backtrace [7] whole read; [8] sides read; [9] reads.
 */
int bt2g_align_profile_read(bt2g_ctx *ctx, uint64_t *out32, int reset, void *stream);

/* ---- index construction ---------------------------------- */
/*
 * Replaces bowtie2-build-{s,l}: the driver (bt2_build.cpp:364-547), KarkkainenBlockwiseSA (blockwise_sa.h) and
 * Ebwt::buildToDisk (bt2_idx.h:2829-3174).  Suffix sorting, BWT/Occ emission, the SA sample and the ftab run on the
 * GPU (radix sort + prefix doubling, bt2g_build_core.hpp); the files <out_base>.{1,2,3,4,rev.1,rev.2}.bt2[l] come out
 * byte-identical to the reference builder's.  Needs no ctx; `device` picks the GPU.  Synchronous.
 */
typedef struct {
	int32_t large_index;      /* 0: .bt2 (32-bit offsets), 1: .bt2l (bowtie2-build --large-index)        */
	int32_t off_rate;         /* -o/--offrate, default 4                                                */
	int32_t ftab_chars;       /* -t/--ftabchars, default 10                                             */
	int32_t write_ref;        /* also write <base>.3 / <base>.4 (default 1; 0 = -r/--noref)              */
	int32_t device;
} bt2g_build_params;
typedef struct {
	uint64_t len, n_pat, n_frag;          /* joined-text length, sequences, N-free fragments              */
	uint32_t rounds_fw, rounds_bw;        /* prefix-doubling rounds per direction                          */
	uint64_t tied_fw, tied_bw;            /* suffixes still tied after the first 29-base sort              */
	double   t_parse, t_fw, t_bw, t_write;/* seconds: input scan, forward index, mirror index, file output  */
} bt2g_build_stats;
void bt2g_build_params_default(bt2g_build_params *p);
/* FASTA files (plain or gzip).  stats may be NULL. */
int bt2g_index_build(const char *const *fasta_paths, uint32_t n_paths, const char *out_base,
                     const bt2g_build_params *params, bt2g_build_stats *stats);
/* Sequences already in host memory (ASCII, any case, IUPAC codes / '-' count as N); names may be NULL ("0","1",...). */
int bt2g_index_build_mem(const char *const *names, const char *const *seqs, const uint64_t *lens, uint32_t n_seqs,
                         const char *out_base, const bt2g_build_params *params, bt2g_build_stats *stats);

/* ---- the whole aligner as one call --------------------------------------- */
/*
 * The reference's library-style entry point, same name and signature: `extern "C" int bowtie(int argc, const char **argv)`
 * (bt2_search.cpp:5223; bowtie_main.cpp:30-67 is the main() around it).  argv is the bowtie2-align-{s,l} command line
 * (argv[0] = program name); SAM goes to -S or stdout, the summary to stderr, the return value is the exit status.
 * bowtie2_amd/bin/bowtie2-align-{s,l} are a main() around this function.  Conditions found before the pipeline starts
 * (bad arguments, no index, no gfx950 device, unreadable files) return non-zero; a malformed record met while the
 * reader/worker threads are running ends the process from that thread, as in the reference.
 */
int bowtie(int argc, const char **argv);
/* Likewise for the index builder: the reference's `extern "C" int bowtie_build(int argc, const char **argv)` (bt2_build.cpp:556-560).
 * argv[0] ending in "build-l" (or --large-index) selects the .bt2l format. */
int bowtie_build(int argc, const char **argv);
/* The executable's option -> parameter mapping (bt2_search.cpp:504-1850 parseOption / parseOptions and the per-read derivations of
 * multiseedSearchWorker, :3341-3450) for callers that drive bt2g_align_batch themselves: argv as bowtie2-align takes it (files may be
 * left out), read_len = length of the reads of the batch.  *params as the drop-in binary would pass them, *rp = per-read parameters of
 * an N-free read of that length (RNG seed left 0: it depends on the read).  both_mates_pass: the x1.2 seed interval of a pair. */
int bt2g_cli_params(int argc, const char **argv, uint32_t read_len, int large_index, int both_mates_pass,
                    bt2g_align_params *params, bt2g_read_params *rp);

/* ---- instrumentation ---------------------------------------------------- */
typedef struct {
	uint64_t rank_queries;    /* # sides read (SURVEY.md 8d unit)             */
	uint64_t sa_lookups;      /* # offs[] reads                               */
	uint64_t ftab_lookups;    /* # ftab jumps                                 */
	uint64_t dp_cells;        /* # DP cells filled                            */
	uint64_t bwops;           /* reference-compatible BW-op count             */
} bt2g_counters;
/* Copies device-side counters accumulated since the last reset (synchronises the stream). */
int bt2g_counters_read(bt2g_ctx *ctx, bt2g_counters *out, int reset, void *stream);

#ifdef __cplusplus
}
#endif
#endif
