// bt2g_align.hpp -- the per-read multiseed worker, written once for the device.
//
// Execution model on MI355X: ONE WAVEFRONT PER READ.  All 64 lanes run this control code in
// lock-step on identical values (it is wave-uniform, so the compiler keeps most of it on the
// scalar unit); the lanes only take different roles inside the explicitly wave-parallel
// sections (the DP fill, the backtrace's tile fetches and diagonal runs, list and table scans).  Per-read working state lives in a per-wave arena in
// HBM (`Work`), sized for 288 GB parts: nothing is allocated dynamically.
//
// What it restates (behaviour, not code) -- cited where each piece starts:
//   multiseedSearchWorker            bt2_search.cpp:3094-4254
//   SeedAligner::exactSweep/oneMmSearch/instantiateSeeds/searchAllSeeds   aligner_seed.cpp
//   SeedResults::rankSeedHits        aligner_seed.h:1019
//   SwDriver::eeSaTups/extend/prioritizeSATupsRands/extendSeeds           aligner_sw_driver.cpp
//   Random1toN, RowSampler           random_util.h, aligner_sw_driver.h:179
//   DynProgFramer::frameSeedExtensionRect   dp_framer.cpp:81
//   SwAligner::initRef/align/nextAlignment/ungappedAlign, e2e u8 gather+backtrace
//                                    aligner_sw.cpp, aligner_swsse_ee_u8.cpp
//   RedundantAlns, EIvalMergeListBinned      aligner_result.cpp:929, ival_list.h
//   AlnSinkWrap::report/finishRead/selectByScore, ReportingState          aln_sink.cpp
//
// Scope: unpaired reads and pairs (bt2g_align_pe.inc), end-to-end and --local, -N 0/1, -M / -k / -a
// reporting, reads up to kMaxLen.  Anything else is rejected up front by the host; a read that outgrows one of the fixed
// capacities below is flagged in its result record.
//
// The same source also compiles for the host (tests/hostsim) -- there the wave-parallel
// sections fall back to plain loops -- which is how the control logic was debugged against the
// reference without a GPU in the build container.  The host build is test-only.
#ifndef BT2G_ALIGN_HPP_
#define BT2G_ALIGN_HPP_

#include "bt2g_device.hpp"
#include "../../include/bt2g.h"

namespace bt2g {

#ifdef BT2G_CLASS_MAX_LEN
// The worker's short-read class (Makefile: bt2g_align_kernel_w5.o -- the same source compiled with 96 registers under another namespace):
// capacities cut to what batches of short unpaired reads need, so that the LDS this frees holds the backtrace's on-chip state
// (DevPlat::rt_begin) at 20 waves per CU.  bt2g_align_batch picks the class per batch and only gives this one batches it holds
// (longest read <= BT2G_CLASS_MAX_LEN, at most BT2G_CLASS_MAX_OFFS seed positions per strand): nothing is flagged that the general class would align.
constexpr int kMaxLen      = BT2G_CLASS_MAX_LEN;
constexpr int kMaxOffs     = BT2G_CLASS_MAX_OFFS;
#else
constexpr int kMaxLen      = 512;   // longest read (DP rows)
constexpr int kMaxOffs     = 64;    // seed offsets per strand
#endif
constexpr int kMaxMm1      = 1024;  // 1-mismatch end-to-end hits kept (a simple-repeat read has hundreds)
constexpr int kMaxRanges   = 2 * kMaxOffs;   // seed positions (both strands)
constexpr int kMaxSat2     = 4096;  // seed-hit ranges of one round: one per position with -N 0, up to ~100 per position with -N 1
#ifdef BT2G_CLASS_MAX_EDITS
constexpr int kMaxEdits    = BT2G_CLASS_MAX_EDITS;      // (the long-read class: its result records have larger alignment slots, see AlnRes below)
#else
constexpr int kMaxEdits    = 200;
#endif
// Edits one backtrace walk may collect before it ends (Edit ned[] in LDS).  A walk that SUCCEEDS must fit an alignment slot (kMaxEdits); a walk of a
// long read that wanders through hundreds of mismatches and then fails must not flag the read.  Local mode only (a local walk stays above score 0, so it
// has at most as many mismatches as matches, plus the Ns the N ceiling admits: (cells + 0.15 L) / 2; end to end the minimum score bounds the edits).
#ifdef BT2G_CLASS_MAX_WALK_EDITS
constexpr int kMaxWalkEdits = BT2G_CLASS_MAX_WALK_EDITS;
#else
constexpr int kMaxWalkEdits = kMaxEdits;
#endif
static_assert(kMaxWalkEdits >= kMaxEdits, "walk buffer smaller than an alignment slot");
#ifdef BT2G_CLASS_BIG_K
// The worker's many-alignments class (Makefile: bt2g_align_kernel_bk.o, namespace bt2g_bk): -k above 64 and -a.  The reference has no ceiling on
// -k (aln_sink.cpp:33-326); this class holds BT2G_MAX_KHITS alignments per read (per mate and per pair list), the extension list that
// maxIters = 400 + 20 (k - 1) rows can fill, and the sampler lists that go with it: 23 MB of arena per wave instead of 7, so bt2g_align_batch gives
// it only the batches that ask for it (their result records are 1.3 MB per read: the driver cuts such batches to ~1 700 reads).
constexpr int kMaxSatpos   = 20608; // maxIters(400 + 20*(k-1), k <= 1000) + ranges + slack
constexpr int kMaxRedAnchor = 4160;
constexpr int kMaxAlnsU    = 2080;
constexpr int kMaxAlns     = 1040;  // BT2G_MAX_KHITS + slack (the sink stops at k; -a flags a read that has more)
constexpr int kMaxDiags    = 8192;
constexpr int kListArena   = 1 << 20;
#else
constexpr int kMaxSatpos   = 1856;  // maxIters(400 + 20*(k-1), k <= 64) + ranges + slack
constexpr int kMaxRedAnchor = 1280;  // alignments remembered by the paired-end redundancy set
constexpr int kMaxAlnsU    = 640;   // unpaired alignments kept per mate of a pair: every distinct opposite-mate alignment found during mate rescue lands here
constexpr int kMaxAlns     = 160;   // alignments kept by the sink (-M 50 -> at most 51; -k <= 64)
constexpr int kMaxDiags    = 2304;  // seen-diagonal intervals
constexpr int kListArena   = 65536; // uint32 slots for Random1toN lists (swap lists of small ranges, seen lists bounded by max_iters, converted lists)
#endif
#ifdef BT2G_CLASS_MAX_CANDS
constexpr int kMaxCands    = BT2G_CLASS_MAX_CANDS;      // (the long-read class: a local window of a 2 000-bp read has hundreds of thousands of candidate cells)
#else
constexpr int kMaxCands    = 65536;  // DP backtrace candidates (>= DP columns; local mode: cells, see gather_local)
#endif
// counters of the local gather's radix sort (DevPlat::gather_local): 1 024 of them, 16 bits wide while every count stays below 65 536
constexpr uint32_t kRadixCntBytes = kMaxCands > 65536 ? 4096u : 2048u;
constexpr int kMaxCols     = 1100;  // DP columns a launch holds unless the caller asks for more: seed extension needs rows + 4*15 + 1; opposite-mate windows span about -X + rows + 2*gaps
// Widest DP window any launch can hold (bt2g_align_params::max_dp_cols asks for it).  The per-column state of the window in flight -- the
// reference masks and the last row's scores -- lives in LDS, and LDS decides how many waves a CU holds: the common batch (unpaired reads,
// pairs with the default -X 500) is launched with kMaxCols columns of it and runs 16 waves per CU; a batch whose opposite-mate windows are
// wider (-X 800, --local pairs of long reads, --dovetail) is launched with as many as it needs, up to this, and runs 12.
constexpr int kMaxColsWide = 2176;

enum { EDIT_READ_GAP = 1, EDIT_REF_GAP = 2, EDIT_MM = 3 };
enum { EXT_EXHAUSTED = 1, EXT_POLICY_FULFILLED, EXT_PERFECT_SCORE, EXT_SOFT_LIMIT, EXT_HARD_LIMIT };
enum { ERR_NONE = 0, ERR_OVERFLOW = 1 };   // per-read status: capacity of a fixed arena exceeded

// ---------------------------------------------------------------------------------------
// The batch parameters, per-read parameters and result records are the C-ABI structs of
// include/bt2g.h (bt2g_align_params, bt2g_read_params, bt2g_edit, bt2g_aln, bt2g_read_result).
using AlignParams = bt2g_align_params;
using ReadParams  = bt2g_read_params;
using Edit        = bt2g_edit;
#ifdef BT2G_CLASS_MAX_EDITS
// the class's own alignment slot and result record: the ABI structs' fields, the slot with room for kMaxEdits edits (include/bt2g.h at
// bt2g_align_result_stride: a record says how large its slots are)
struct AlnRes {
	int64_t  refoff, reflen;
	int32_t  refid, score;
	int16_t  ns, gaps, edits, bases_aligned;
	uint16_t refns, nned, rdlen, rdextent, rfextent, trim5p, trim3p;
	uint8_t  fw;
	uint8_t  pad[5];
	bt2g_edit ned[kMaxEdits];
};
struct ReadResult {
	uint8_t  status;
	uint8_t  aligned, maxed, filt, exhausted, has_secbest;
	uint8_t  pair_type;
	uint8_t  pair_flags;
	int32_t  secbest, best;
	uint32_t nalns, nreport;
	uint32_t n_ex_iters, n_ex_dps, n_ex_ugs, n_dp_fail_streak_max, n_bwops_seed, n_bwops_ext, n_redundants, n_bt_attempts;
	uint32_t n_ext_left, n_ext_right, n_resolve_steps, n_sides;
	int32_t  pair_best, pair_secbest;
	uint32_t n_mate_dps, pad2;
	AlnRes   alns[1];
};
static_assert(offsetof(AlnRes, ned) == offsetof(bt2g_aln, ned) && offsetof(AlnRes, nned) == offsetof(bt2g_aln, nned) && offsetof(AlnRes, fw) == offsetof(bt2g_aln, fw) &&
              offsetof(ReadResult, alns) == offsetof(bt2g_read_result, alns) && offsetof(ReadResult, pad2) == offsetof(bt2g_read_result, pad2) &&
              offsetof(ReadResult, nreport) == offsetof(bt2g_read_result, nreport) && kMaxEdits <= BT2G_MAX_EDITS_LONG && sizeof(AlnRes) % 8 == 0, "class record layout");
#else
using AlnRes      = bt2g_aln;
using ReadResult  = bt2g_read_result;
#endif
// what a record says about its alignment slots (bt2g_read_result::pad2 bits 16-31)
constexpr uint32_t kAlnSlotTag = (uint32_t)(sizeof(AlnRes) / 8) << 16;
static_assert(kMaxLen <= ((BT2G_MAX_READ_LEN + 63) & ~63) && kMaxLen % 4 == 0 && (kMaxEdits == BT2G_MAX_EDITS || kMaxEdits == BT2G_MAX_EDITS_LONG), "ABI constants out of sync");

// ---------------------------------------------------------------------------------------
// RandomSource (random_source.h:34-159)
struct Rng {
	uint32_t last, lastOff;
	BT2_HD void init(uint32_t seed) { last = seed; lastOff = 30; }
	BT2_HD uint32_t nextU32() {
		last = 1664525u * last + 1013904223u;
		uint32_t ret = last >> 16;
		last = 1664525u * last + 1013904223u;
		ret ^= last;
		lastOff = 0;
		return ret;
	}
	BT2_HD uint64_t nextU64() { uint64_t hi = nextU32(); uint64_t lo = nextU32(); return (hi << 32) | lo; }
	BT2_HD bool nextBool() {
		if (lastOff > 31) nextU32();
		const uint32_t r = (last >> lastOff) & 1u;
		lastOff++;
		return r != 0;
	}
	BT2_HD float nextFloat() { return (float)nextU32() / (float)0xffffffffu; }
};

// ---------------------------------------------------------------------------------------
// working state of one read (one wavefront)
struct EEHit {            // end-to-end exact / 1-mismatch hit (aligner_seed.h:482)
	uint64_t top, bot;
	int32_t  score;
	uint16_t epos;        // mismatch: offset from 5' end
	uint8_t  echr, eqchr; // mismatch: reference char / read char (codes 0..4)
	uint8_t  fw, has_edit;
	uint8_t  pad[2];
};

struct SeedHitRec {       // one (strand, offidx): QVal with a single range (exact seeds)
	uint64_t topf, botf, topb, botb;   // botf==topf: no hit
};

struct R1N {              // Random1toN (random_util.h:32)
	uint32_t sz, n, cur, thresh;
	uint32_t list_off, list_len;   // into Work::lists
	uint32_t seen_off, seen_len;
	uint8_t  swaplist, converted, inited, pad;
};

struct SatPos {           // SATupleAndPos (aligner_sw_driver.h:144)
	uint64_t topf, topb;
	uint32_t size;        // sat.size() (elements in this tuple)
	uint32_t orig_sz;
	uint32_t offidx, rdoff, seedlen;
	uint32_t nlex, nrex;
	uint8_t  fw, pad[3];
	int32_t  ee;          // index into eehits (eeMode) or -1
	R1N      rnd;         // rands_[i]
};

// A row drawn by the RowSampler: which range (Work::satpos2 index) and which row.  prioritize() draws up to max_iters of
// them per call but the extension loop consumes a few dozen, so only the consumed ones are expanded to a SatPos.
struct SampRow { uint64_t topf; uint32_t src; uint32_t done; };

// RowSampler::init's weight of a range (aligner_sw_driver.h:176-200, lensq = szsq = true)
BT2_HD double samp_mass(uint32_t nlex, uint32_t nrex, uint32_t size) {
	double num = (double)(nlex + nrex + 1); num *= num;
	double denom = (double)size; denom *= denom;
	return num / denom;
}
BT2_HD double f64_of(uint32_t lo, uint32_t hi) { const uint64_t u = (uint64_t)lo | ((uint64_t)hi << 32); double d; __builtin_memcpy(&d, &u, 8); return d; }

struct DiagIval { int64_t off; int64_t len; int32_t ref; int32_t orient; };

struct BtCand { int32_t score; uint16_t row, col; };
constexpr int kMaxLocalScore = 2047;      // counting-sort table of the local candidate gather (scores above share the top bucket)
constexpr int kMaxCandDone = 1024;       // local mode: candidates of one window that can be tried (btncanddone_) before the read is flagged

struct DPRect { int64_t refl, refr, refl_pretrim, refr_pretrim; uint32_t triml, trimr, corel, corer, maxgap; };

// The hot, small part of the per-read state.  On the device it lives in LDS (one wavefront per
// workgroup): these arrays are touched by almost every step of the scalar control code, and an
// LDS access costs a few issue cycles where a wave-uniform global load occupies the vector
// memory pipeline for a full 64-lane address pass.
// ReportingState of a pair (aln_sink.h:380-580) and the running bests of AlnSinkWrap (aln_sink.cpp:1395-1452)
enum { PE_EXIT_DID_NOT_ENTER = 1, PE_EXIT_DID_NOT_EXIT, PE_EXIT_K, PE_EXIT_M, PE_EXIT_TRUMPED, PE_EXIT_CONVERTED, PE_EXIT_NO_ALNS, PE_EXIT_WITH_ALNS };
struct PeHot {
	uint8_t  cur;                       // mate whose read and seed state are loaded
	uint8_t  done_concord, done_discord, done_unp[2], done_all;
	uint8_t  exit_concord, exit_discord, exit_unp[2];
	uint32_t n_concord, n_discord, n_unp[2];
	int64_t  best_pair, best2_pair, best_unp[2], best2_unp[2];
	uint32_t n_red_anchor;
	uint32_t n_ex_fw2, n_ex_rc2;        // mate 2's covered-range lists (HotWork::n_ex_fw/rc are mate 1's)
	uint32_t n_mate_dps, n_mate_ugs;
	uint32_t olen;                      // length of the mate that is not loaded
};
// The reference keeps the seed hits of one seeding round in an AlignmentCache whose memory is a fixed pool of 16 KB pages
// (--seed-cache-sz, 20 MB: aligner_cache.h:66,466-480, ds.h:3067-3150).  Every SA range is stored with one pool slot PER
// ELEMENT (AlignmentCache::addOnTheFlyImpl, aligner_cache.cpp:53-105), so a read whose seeds hit millions of rows --
// simple repeats, the commonest Alu words -- exhausts the pool: the range being stored is cut to what fitted, the seed
// that was being added is dropped (SeedAligner::searchAllSeeds counts an "oom" and skips sr.add, aligner_seed.cpp:672-690),
// later seeds with a new sequence are dropped too, and later seeds with an already-stored sequence see the cut range.
// SAM parity needs exactly that, so the worker replays the pool accounting per seeding round: pages handed to the QKey map,
// the SAKey list, the SAKey map and the element list, in seed order (fw offsets, then rc offsets).
constexpr int kCacheKeys = 2 * kMaxOffs * 2 + kMaxSat2;      // distinct seed sequences of one round (both mates of a pair) + with -N 1 the distinct reference strings they hit
struct CacheModel {
	uint32_t pool_total, pool_used;
	uint32_t qn, ql, san;          // nodes in the QKey map, entries in the SAKey list, nodes in the SAKey map
	uint64_t sl;                   // elements in the element list
	uint32_t nkeys;
	uint32_t fast_round;           // pairs: the round's ranges of BOTH mates provably fit the pool (cache_filter): neither mate's seeds go through the model
};

// `size` = elements the seed search found (what SeedResults tallies and ranks by); `esize` = elements of the range as the
// per-read seed cache holds it: smaller when the cache's page pool ran out while the range was stored (struct CacheModel)
struct HotHit { uint64_t topf, topb; uint32_t size; uint32_t esize; };   // exact seed hit: one range.  With -N 1: topf = first entry in Work::sranges, topb = # ranges, size = total elements
struct SeedRange { uint64_t topf, topb; uint32_t size; uint32_t esize; };   // one BW range of a 1-mismatch seed (SATuple, aligner_cache.h:370); esize: elements the seed cache holds for its reference string (CacheModel)
struct HotWork {
	uint8_t  seq[kMaxLen];     // read, codes 0..4, 5'->3'
	uint8_t  qual[kMaxLen];    // ASCII
	// (the reference masks of the current DP window, the edits of the backtrace in progress and the scores of the last DP row live behind
	// Plat::rf() / ned() / lastrow(): sized per launch, see kMaxColsWide)
	HotHit   hits[2][kMaxOffs];// [0] fw seeds, [1] rc seeds; size==0 => no hit
	uint8_t  sorted[2][kMaxOffs];
	uint8_t  rank_offs[kMaxRanges];
	uint8_t  rank_fw[kMaxRanges];
	// ---- scalar control state of the read in flight (everything the control code touches often) ----
	uint32_t len;
	EEHit    exact[2];         // [0] fw, [1] rc; top==bot => empty
	uint32_t n_mm1;
	uint64_t mm1_elt;
	uint32_t num_offs;
	uint16_t off_idx2off[kMaxOffs];   // read offsets of the seed positions (< kMaxLen)
	uint32_t n_rank;
	uint32_t nonz_tot, nonz_fw, nonz_rc;
	uint64_t num_elts;
	uint32_t n_satpos2;
	uint32_t n_satpos;
	uint32_t n_resolved;        // sampled rows [n_satpos_full, n_resolved) have their offset in Work::srow_joff
	uint32_t n_satpos_full;     // entries [0, n_satpos_full) of the extension list are whole SatPos records, the rest sampled rows (Work::srows)
	uint32_t lists_used;
	double   mass;
	uint32_t n_masses;
	uint32_t samp_sai, samp_lanes;      // first range of the row sampler in Work::satpos2; 1: at most 64 ranges (their fields fit one lane register, extend_seeds)
	uint32_t n_ex_fw, n_ex_rc;
	uint32_t n_diags;
	uint32_t n_alns;
	int64_t  best_unp1, best2_unp1;
	uint8_t  done_unpair1;
	uint8_t  exit_m, exit_k;
	uint32_t n_cands, cural;
	uint32_t n_cdone;           // local mode: candidates of the window in hand that have been tried (btncanddone_), listed in Work::cand_done
	uint32_t err;
	uint32_t n_ex_iters, n_ex_dps, n_ex_ugs, n_dp_fail, n_ug_fail, n_ee_fail, n_dp_fail_streak;
	uint32_t n_redundants, n_bwops_seed, n_bwops_ext, n_bt_attempts;
	uint32_t n_ext_left, n_ext_right, n_resolve_steps;
	uint32_t n_sides;           // sides (64/128-byte lines) actually read -- roofline accounting
	uint32_t n_sranges;         // -N 1: entries of Work::sranges in use
	// the N-free fragment of the reference (rstarts record) the last resolved seed hit lies in: joined-text start, length, offset of its
	// first base within reference frag_tidx.  A DP window inside it is a contiguous piece of the joined text (= the .4 buffer).
	uint64_t frag_jlo, frag_len, frag_toff, frag_tidx;
	uint32_t n_dp_cells_score, n_dp_cells_full, n_dp_pass;   // measurement: DP cells computed by score-only passes / by fills that store a matrix; windows that reached minsc
	CacheModel cm;              // seed cache pool of the current seeding round
	PeHot    pe;                // paired-end reporting state (unused for unpaired reads)
	uint64_t t_bt[5];           // profile of the backtraces: ticks in all walks, in successful walks, # successful, ticks after the trace of a successful walk, scalar steps
	uint64_t t_phase[22];       // device clock ticks per phase (profiling): 0 sweep 1 mm1 2 seeds 3 rank+prioritise 4 resolve 5 dp fill 6 gather+backtrace 7 other
};
// Backtrace tile: the cells a run of kBtTile diagonal steps starting at (row, col) can look at, gathered with one
// lane-parallel load into a per-lane register (Plat::bt_tile): lanes 0-15 cell(row-d, col-d), 16-31 cell(row-d-1, col-d),
// 32-47 cell(row-d, col-d-1), 48-63 mask(row-d, col-d), d = lane & 15.
constexpr uint32_t kBtTile = 15;

struct Work {
	// per-round seed cache model (CacheModel): one entry per distinct seed sequence
	uint64_t ck_key[kCacheKeys];
	uint32_t ck_eff[kCacheKeys];   // elements the stored range holds (after a possible cut)
	uint8_t  ck_len[kCacheKeys];
	uint8_t  ck_flags[kCacheKeys]; // bit 0: in the QKey map, bit 1: in the SAKey map
	// ---- read ----
	// ---- seed phase ----
	EEHit    mm1[kMaxMm1];
	// ---- extension phase ----
	SeedRange sranges[kMaxSat2];       // -N 1: the ranges behind HotWork::hits
	uint64_t srange_key[kMaxSat2];     // -N 1: the reference string of each range (the seed with its substitution), packed like ck_key
	SatPos   satpos2[kMaxSat2];
	R1N      rands2[kMaxSat2];
	SatPos   satpos[kMaxSatpos];
	SampRow  srows[kMaxSatpos];
	uint64_t srow_joff[kMaxSatpos];     // their text offsets (joff_pack), resolved 64 rows at a time by all lanes
	SatPos   sp_view;                   // the sampled row being extended, expanded
	uint32_t lists[kListArena];
	double   masses[kMaxSat2];
	uint8_t  elim[kMaxSat2];
	struct ExtRange { uint32_t off, len, sz; } ex_fw[kMaxRanges * 2], ex_rc[kMaxRanges * 2];   // seedExRangeFw_/Rc_[0]
	ExtRange ex_fw2[kMaxRanges * 2], ex_rc2[kMaxRanges * 2];                                   // ... [1]: mate 2 of a pair
	DiagIval diags[kMaxDiags];
	int64_t  red_dmin[kMaxAlns], red_dmax[kMaxAlns];   // RedundantAlns prefilter: (column - row) bounds of alns[k]
	// ---- sink ----
	AlnRes   alns[kMaxAlns];
	// ---- DP ----
	BtCand   cands[kMaxCands];
	uint32_t cand_hist[2 * (kMaxLocalScore + 1)];   // scratch of the local gather's counting sort
	uint32_t cand_done[2][kMaxCandDone];            // local mode: tried candidates (row | col << 16) of the anchor's window and of the opposite mate's
	AlnRes   res;                       // resGap_ / resEe_ / resUngap_
	BtCand   cands2[kMaxCands];         // candidates of the opposite-mate DP (the anchor's stay live in `cands`)
	BtCand   cands_tmp[kMaxCands];      // device, local mode: the candidate cells as the fill meets them; the gather sorts them into cands / cands2
#ifndef BT2G_NO_PAIRS      // (the short-read class aligns unpaired end-to-end batches only: its work area ends here -- 0.8 MB instead of 7 MB per wave, one or two
	                       // 2-MB pages of address space per resident wave instead of four or five, see align_scratch_sizes)
	// ---- paired-end (extendSeedsPaired; unused for unpaired reads) ----
	// seed-phase state of the mate that is not loaded (both mates are searched before either is extended)
	struct MateSave {
		uint32_t len;
		EEHit    exact[2];
		uint32_t n_mm1; uint64_t mm1_elt;
		uint32_t num_offs, nonz_tot, nonz_fw, nonz_rc, n_sranges; uint64_t num_elts;
		uint32_t off_idx2off[kMaxOffs];
		HotHit   hits[2][kMaxOffs];
		EEHit    mm1[kMaxMm1];
		SeedRange sranges[kMaxSat2];
	} ms[2];
	AlnRes   alns_u[2][kMaxAlnsU];       // rs1u_ / rs2u_: unpaired alignments per mate (also redMate1_/redMate2_)
	AlnRes   alns_p[2][kMaxAlns];       // rs1_ / rs2_: concordant (or the one discordant) pair, same index
	int64_t  redu_dmin[2][kMaxAlnsU], redu_dmax[2][kMaxAlnsU];
	AlnRes   red_anchor[kMaxRedAnchor]; // redAnchor_: every alignment found for either mate while it was the anchor or the rescued mate
	int64_t  reda_dmin[kMaxRedAnchor], reda_dmax[kMaxRedAnchor];
	AlnRes   ores;                      // oresGap_
	uint32_t mate_streaks[kMaxSatpos];  // mateStreaks_
#endif
	// ---- status / metrics ----
};

// DP scratch addressing (wavefront-major, see bt2g_kernels.hip)
// rows per lane of the wavefront-major 16-bit matrix; the long-read class rounds up to the fills it instantiates (8 and below as they are, then 16, 24, 32)
BT2_HD uint32_t dp_R(uint32_t rows) {
	const uint32_t r = (rows + 63) / 64;
	return (kMaxLen > 512 && r > 8) ? (r + 7) & ~7u : r;
}
// wavefront-major, one packed word per cell: H | E<<8 | F<<16 at word index (t*R + r)*64 + lane
BT2_HD uint64_t dp_cell(uint32_t R, uint32_t i, uint32_t j) {
	const uint32_t l = i / R, r = i % R;
	const uint64_t t = (uint64_t)j + l;
	return (t * R + r) * 64 + l;
}
// Local fills compute two cells per register (bt2g_local_pk.hpp): the read is cut into blocks of RB rows, block k belongs to lane k & 63 (low
// halves for k < 64, high halves above) and meets column j in step t = j + k.  One byte per cell; the 64 low-half lanes' bytes of a (step, row in
// block) are consecutive, then the 64 high-half ones: every store instruction of the fill writes 64 consecutive bytes.
BT2_HD uint32_t dp_RB(uint32_t rows) { const uint32_t r = (rows + 127) / 128; return r <= 4 ? r : r <= 8 ? 8u : 16u; }      // (what the packed fill is instantiated for: 1..4 up to 512 rows; 8 and 16 in the long-read class)
BT2_HD uint64_t dp_cell_pk(uint32_t RB, uint32_t i, uint32_t j) {
	const uint32_t k = i / RB, r = i % RB;
	const uint64_t t = (uint64_t)j + k;
	return (t * RB + r) * 128 + (k >> 6) * 64 + (k & 63);
}
// bytes of the anti-diagonal matrix of a rows x cols local problem
BT2_HD uint64_t dp_pk_cells(uint32_t rows, uint32_t cols) { const uint32_t RB = dp_RB(rows); return ((uint64_t)cols + (rows + RB - 1) / RB) * RB * 128; }


// DP scratch of one wave.  Two matrix formats:
//  * end-to-end 8-bit mode (the common case): ONE BYTE per cell holding which predecessors are score-consistent
//    (PB_* below), computed while the cell is filled -- the backtrace never looks at scores again.  Only the band of
//    diagonals a valid alignment can touch exists (EeBand / pred_idx), row-major over diagonals; the backtrace gathers 64
//    diagonal steps per fetch.  Its per-cell backtrace masks (`pmask`, 32 bits) carry an epoch tag instead of being cleared
//    for every DP;
//  * local mode: the same byte per cell (with the local kernels' `> floor` rule folded in: a predecessor whose score is 0 is none), over
//    the whole rectangle, in anti-diagonal order (pred_at);
//  * 16-bit end-to-end: wavefront-major packed H|E|F cells (dp_cell) + a 16-bit mask plane that is zeroed
//    after every fill that has candidate cells.
struct DpScratch {
	BT2_G uint32_t* mat;      // pred bytes (8-bit end-to-end) or packed cells
	BT2_G uint16_t* masks;    // [rows][cols] masks of the packed-cell formats
	BT2_G uint32_t* pmask;    // masks of the pred format: bits 0-12 as SSEMatrix::masks_, bits 13-31 = epoch of the DP that wrote them
	BT2_G uint32_t* epoch;    // -> [0] epoch of the DP currently in this scratch (lives in the arena, survives launches), [1] band lo, [2] band row width
	uint32_t  pmask_words;
};
// predecessor bits of one cell (aligner_swsse_ee_u8.cpp:1330-1520 asks these questions during the backtrace):
//  HD: H came diagonally;  HE / HF: H equals E / F of the cell and gaps are allowed in this row;
//  EO / EE: E opens from H-left / extends E-left;  FO / FE: F opens from H-up / extends F-up
enum { PB_HD = 1, PB_HE = 2, PB_HF = 4, PB_EO = 8, PB_EE = 16, PB_FO = 32, PB_FE = 64 };
constexpr uint32_t kEpochShift = 13, kEpochMax = (1u << 19) - 1;
constexpr uint32_t kPredTile = 64;   // diagonal steps one tile fetch covers
// Geometry of the pred format.  Only a BAND of diagonals of the DP rectangle is ever filled: an end-to-end alignment that
// scores >= minsc starts in row 0 at a column >= 0, ends in the last row at a column < cols, and makes at most G vertical
// moves (reference gaps: the first costs rfgapo, each further one rfgape, and the perfect score is 0), so every cell (i, j)
// on such a path has  -G <= j - i <= cols - rows + G.  Cells outside the band can only lie on paths that score < minsc;
// treating them as "minus infinity" changes neither the last-row scores >= minsc nor any predecessor bit of a cell the
// backtrace can visit (a predecessor that is consistent with a visited cell lies on a valid path itself, and a valid path
// never leaves the band).  Cell (i, j) is stored at byte i * w + (j - i + lo): row-major over diagonals, one lane of the
// fill owns 2 * RP consecutive diagonals (w = 128 * RP).  lo/w live in the scratch header next to the epoch.
struct EeBand { int32_t lo; uint32_t nd; };      // diagonal index dd = j - i + lo, 0 <= dd < nd
BT2_HD bool ee_band(int rfgapo, int rfgape, uint32_t rows, uint32_t cols, int64_t minsc, EeBand& b) {
	const int64_t budget = -minsc;
	if (budget < 0 || rows == 0 || cols == 0) return false;
	int64_t g;
	if (budget < rfgapo) g = 0; else if (rfgape <= 0) g = rows; else g = (budget - rfgapo) / rfgape + 1;
	if (g > (int64_t)rows - 1) g = (int64_t)rows - 1;
	int64_t dhi = (int64_t)cols - (int64_t)rows + g;
	if (dhi > (int64_t)cols - 1) dhi = (int64_t)cols - 1;
	if (dhi < -g) return false;
	b.lo = (int32_t)g; b.nd = (uint32_t)(dhi + g + 1);
	return true;
}
// pairs of diagonals per lane the fill is instantiated for (0: more than 2048 diagonals -- cannot happen within kMaxLen / kMaxCols)
BT2_HD uint32_t ee_band_rp(uint32_t nd) {
	const uint32_t need = (nd + 127) / 128;
	return need <= 4 ? need : need <= 6 ? 6u : need <= 8 ? 8u : need <= 12 ? 12u : need <= 16 ? 16u : 0u;
}
BT2_HD uint64_t pred_idx(int32_t lo, uint32_t w, uint32_t i, uint32_t j) { return (uint64_t)i * w + (uint32_t)((int32_t)j - (int32_t)i + lo); }
// The predecessor-byte matrix of a LOCAL fill covers the whole rectangle and is stored in the order its anti-diagonal fill produces it
// (dp_cell_pk: block of RB rows = half a lane, step = column + block).  The scratch header says which form a matrix has: row width w > 0 = band
// form with first diagonal lo; w == 0 = anti-diagonal form with lo = rows per block.
BT2_HD uint64_t pred_at(int32_t lo, uint32_t w, uint32_t i, uint32_t j) { return w ? pred_idx(lo, w, i, j) : dp_cell_pk((uint32_t)lo, i, j); }
// bytes of the widest band a rows x cols problem can have (every diagonal of the rectangle)
BT2_HD uint64_t pred_cells(uint32_t rows, uint32_t cols) { uint32_t rp = ee_band_rp(rows + cols); if (rp == 0) rp = 16; return (uint64_t)rows * rp * 128; }

// The worker's own state (what used to be the data members of Aligner).  On the device ONE instance per wavefront lives in LDS and is
// reached by name (Plat::st()), never through a pointer: see the note at ST in bt2g_align_core.hpp.
struct AlState {
	BT2_G Work* wp;          // the wave's work area in HBM
	BT2_HD Work* wp_generic() const { return (Work*)wp; }
	DpScratch dp;            // the DP scratch in use (pairs: dp_main or dp_opp)
	Rng       rnd;
	int64_t   minsc;         // current (possibly tightened) minimum score
	uint32_t  ridx;          // index of this read in the batch
	bool      ext_pre;       // HOT.hits came from pre->seeds, so pre->ext holds their extensions
	const BT2_G uint32_t* pre_ext_cur;      // extension / resolved-offset tables of the seed round in HOT.hits
	const BT2_G uint64_t* pre_joff_cur;
	uint32_t  pf_steps, pf_tiles;     // profile: backtrace steps / tile fetches of this read
	uint64_t  pf_tile_t;
	uint8_t   m_nofw, m_norc;         // --nofw / --norc as they apply to the loaded read (mate 2 of an --fr pair sees them swapped)
	// ---- pairs (bt2g_align_pe.inc); inputs set by the launcher before run_pair ----
	const BT2_G uint8_t* pe_seq[2];
	const BT2_G uint8_t* pe_qual[2];
	uint32_t  pe_len[2];
	ReadParams pe_rp[2];
	int64_t   pe_minsc[2];
	DpScratch dp_main, dp_opp;        // anchor / opposite-mate matrices (dp selects the one in use)
	BT2_G BtCand* cands_cur;          // candidate list in use (Work::cands, or Work::cands2 during an opposite-mate DP)
	uint32_t  pe_streak;
	uint32_t  pe_pair;                // index of the pair in the batch (reads 2*pe_pair, 2*pe_pair + 1)
	uint32_t  n_emit;                 // device, local mode: candidate cells the last fill wrote to Work::cands_tmp (unsorted; may exceed its capacity)
	int32_t   emit_vmax;              //   ... and the largest score among them
	uint32_t  emit_on;                // 1 in the workers (Aligner's constructor); 0 in the stage kernel, whose waves have no work area
	uint32_t  max_cols;               // DP columns this launch holds (kMaxCols .. kMaxColsWide): wider windows flag the read
	uint32_t  rf_at, ned_at;          // device: LDS addresses of rf and of ned / lastrow in the launch's dynamic LDS (set by the kernel, which knows where that starts)
	// device: the reportedThrough plane of the band matrix in hand, in the launch's dynamic LDS when it has room (DevPlat::rt_begin):
	// LDS address, capacity in bytes, "this matrix's marks are on chip"
	uint32_t  rt_at, rt_bytes, rt_cur;
	uint32_t  wide_cells;             // device, the fill stage only: 1 = a 16-bit end-to-end problem is filled in the anti-diagonal cell form even where the band form fits (BT2G_DP_EE_I16)
	uint32_t  dyn_bytes;              // device: bytes of dynamic LDS this launch has from rf_at on (tail + marks): between DP windows the row sampler keeps its hash tables there
	uint32_t  fill_rows_done, fill_lastsol, fill_sat8;   // device: what a leaf fill hands back besides its return value (row the score-only pass stopped in; lastsolcol_ / "8-bit kernel saturated" of a local fill)
};

// The per-column tail of the hot state, as one launch lays it out behind the fixed part:  rf[max_cols + 8]  (rounded up to 16 bytes), then a
// region shared by  Edit ned[kMaxWalkEdits]  and  int16_t lastrow[max_cols + 8]  (never live at the same time: the gather reads `lastrow` before
// any backtrace writes `ned`; the local gather's radix sort borrows kRadixCntBytes of it for its counters: local launches keep at least that).
// DP columns a launch with these parameters holds (bt2g_align_params::max_dp_cols)
BT2_HD uint32_t dp_cols_for(const AlignParams& P) { return P.max_dp_cols > kMaxCols ? (uint32_t)(P.max_dp_cols < kMaxColsWide ? P.max_dp_cols : kMaxColsWide) : (uint32_t)kMaxCols; }
BT2_HD uint32_t hot_tail_off(uint32_t max_cols) { return (max_cols + 8 + 15) & ~15u; }
BT2_HD uint32_t hot_tail_bytes(uint32_t max_cols, bool local = true) {
	uint32_t b = (max_cols + 8) * 2;
	if (b < (uint32_t)(kMaxEdits * sizeof(Edit))) b = (uint32_t)(kMaxEdits * sizeof(Edit));
	if (kMaxWalkEdits > kMaxEdits && local && b < (uint32_t)(kMaxWalkEdits * sizeof(Edit))) b = (uint32_t)(kMaxWalkEdits * sizeof(Edit));      // (local walks, see kMaxWalkEdits)
	if (local && b < kRadixCntBytes) b = kRadixCntBytes;      // (the radix counters of the local gather)
	return hot_tail_off(max_cols) + ((b + 15) & ~15u);
}

// ---------------------------------------------------------------------------------------
// small helpers
BT2_HD int comp4(int c) { return c < 4 ? 3 - c : 4; }
BT2_HD int imin(int a, int b) { return a < b ? a : b; }
BT2_HD int imax(int a, int b) { return a > b ? a : b; }
BT2_HD int subs0(int a, int b) { const int r = a - b; return r < 0 ? 0 : r; }

// "ACGTN"[c] without a table in memory
BT2_HD uint8_t code2chr(int c) { return (uint8_t)((0x4E54474341ull >> (8 * (c > 4 ? 4 : c))) & 0xff); }

template <typename SP> BT2_HD int mm_penalty(const SP& P, int q) {
	if (P.mm_type == 3) {     // COST_MODEL_QUAL (scoring.h:106-114)
		const int qq = q < 40 ? q : 40;
		const float frac = (float)qq / 40.0f;
		return P.mm_min + (int)(frac * (float)(P.mm_max - P.mm_min));
	}
	if (P.mm_type == 2) return q < 5 ? 0 : (q < 15 ? 10 : (q < 25 ? 20 : 30));   // COST_MODEL_ROUNDED_QUAL: qualRounds[] (qual.cpp:23)
	return P.mm_max;
}

// Scoring::score(rdc, refmask, q) (scoring.h:241)
template <typename SP> BT2_HD int sc_score(const SP& P, int rdc, int refm, int q) {
	if (q < 0) q = 0;
	if (q > 255) q = 255;
	if (rdc > 3 || refm > 15) return -P.n_pen;
	if (refm & (1 << rdc)) return P.match_bonus;
	return -mm_penalty(P, q);
}
// Scoring::mm(rdc, refm, q) (scoring.h:231)
template <typename SP> BT2_HD int sc_mm(const SP& P, int rdc, int refm, int q) {
	if (q < 0) q = 0;
	if (q > 255) q = 255;
	return (rdc > 3 || refm > 15) ? P.n_pen : mm_penalty(P, q);
}

// Scoring::maxReadGaps / maxRefGaps (scoring.cpp:42,73), monotone (match bonus 0 in e2e)
BT2_HD int max_read_gaps(const AlignParams& P, int64_t minsc, uint32_t rdlen) {
	int64_t sc = (int64_t)rdlen * P.match_bonus;
	bool first = true;
	int num = 0;
	while (sc >= minsc) {
		if (first) { first = false; sc -= P.rdgapo; } else sc -= P.rdgape;
		num++;
	}
	return num - 1;
}
BT2_HD int max_ref_gaps(const AlignParams& P, int64_t minsc, uint32_t rdlen) {
	int64_t sc = (int64_t)rdlen * P.match_bonus;
	bool first = true;
	int num = 0;
	while (sc >= minsc) {
		sc -= P.match_bonus;
		if (first) { first = false; sc -= P.rfgapo; } else sc -= P.rfgape;
		num++;
	}
	return num - 1;
}

// Upper bound on the columns (+ 1 padding column) of the DP window in which a mate of length `ordlen` with minimum score `ominsc` is looked
// for next to its aligned partner (extend_seeds_paired: pe_other_mate + pe_frame_mate_rect = otherMate, pe.cpp:237-420, and
// frameFindMateRect, dp_framer.cpp:177-305): the span of admissible fragment ends, the mate itself, and the gap allowance on both sides.
// The caller of bt2g_align_batch takes the maximum over the pairs of a batch for bt2g_align_params::max_dp_cols.
BT2_HD uint32_t mate_window_bound(const AlignParams& P, int64_t ominsc, uint32_t ordlen, uint32_t len1, uint32_t len2) {
	int64_t maxfrag = P.pe_maxfrag;
	const int64_t minfrag = P.pe_minfrag < 1 ? 1 : P.pe_minfrag;
	if ((int64_t)len1 > maxfrag) maxfrag = (int64_t)len1;
	if ((int64_t)len2 > maxfrag) maxfrag = (int64_t)len2;
	int maxgap = max_read_gaps(P, ominsc, ordlen);
	const int rfg = max_ref_gaps(P, ominsc, ordlen);
	if (rfg > maxgap) maxgap = rfg;
	if (maxgap < P.maxhalf) maxgap = P.maxhalf;
	if (maxgap < 0) maxgap = 0;
	const int64_t w = (maxfrag - minfrag) + (int64_t)ordlen + 2 * (int64_t)maxgap + 2;
	return w < 1 ? 1u : w > 0x7fffffff ? 0x7fffffffu : (uint32_t)w;
}

// read accessors: patFw / patRc / qual / qualRev (read.h:73-128)
BT2_HD int rd_char(const HotWork& h, uint32_t len, bool fw, uint32_t i) { return fw ? h.seq[i] : comp4(h.seq[len - 1 - i]); }
BT2_HD int rd_qual(const HotWork& h, uint32_t len, bool fw, uint32_t i) { return fw ? h.qual[i] : h.qual[len - 1 - i]; }

} // namespace bt2g
#endif
