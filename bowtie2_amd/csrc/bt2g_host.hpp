// bt2g_host.hpp -- host side of the drop-in: option handling, read ingest, per-read parameter
// derivation and SAM emission.  None of this is on the GPU hot path; it exists so that the
// per-read worker can be swapped under the reference's bowtie2-align-{s,l} contract
// (argv in, SAM + stderr summary out).
//
// Reference behaviour restated here:
//   option defaults / presets      bt2_search.cpp:303-502, presets.cpp:33-91, aligner_seed_policy.cpp:252-310
//   SimpleFunc::f                  simple_func.h:89-112
//   FASTQ parse, genRandSeed       pat.cpp:45-84,1136-1247
//   filters, minsc, interval       bt2_search.cpp:3352-3450
//   SAM record / flags / MAPQ      aln_sink.cpp:1889-2124, sam.cpp:30-420, unique.h:170-330,
//                                  aligner_result.cpp:556-870 (StackedAln)
//   alignment summary              aln_sink.cpp:349-560
#ifndef BT2G_HOST_HPP_
#define BT2G_HOST_HPP_

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "bt2g_align.hpp"

namespace bt2g {

struct SimpleFunc {
	int type = 0;   // 1 const, 2 linear, 3 sqrt, 4 log
	double I = 0, X = 0, C = 0, L = 0;
	void init(int t, double c, double l) {
		type = t; C = c; L = l; I = -std::numeric_limits<double>::max(); X = std::numeric_limits<double>::max();
	}
	void init(int t, double i, double x, double c, double l) { type = t; I = i; X = x; C = c; L = l; }
	template <typename T> T f(double x) const {
		double v = 0.0;
		if (type == 2) v = x; else if (type == 3) v = std::sqrt(x); else if (type == 4) v = std::log(x);
		const double ret = std::max(I, std::min(X, C + L * v));
		if (ret == std::numeric_limits<double>::max()) return std::numeric_limits<T>::max();
		if (ret == std::numeric_limits<double>::min()) return std::numeric_limits<T>::min();
		return (T)ret;
	}
	// "L,-0.6,-0.6" style (SimpleFunc::parse)
	bool parse(const std::string& s) {
		std::vector<std::string> t;
		size_t p = 0;
		while (true) { size_t q = s.find(',', p); t.push_back(s.substr(p, q == std::string::npos ? q : q - p)); if (q == std::string::npos) break; p = q + 1; }
		if (t.empty() || t[0].empty()) return false;
		const char c = t[0][0];
		if (c == 'C') type = 1; else if (c == 'L') type = 2; else if (c == 'S') type = 3; else if (c == 'G') type = 4; else return false;
		if (t.size() >= 2) C = atof(t[1].c_str());
		if (t.size() >= 3) L = atof(t[2].c_str());
		if (t.size() >= 4) I = atof(t[3].c_str());
		if (t.size() >= 5) X = atof(t[4].c_str());
		return true;
	}
};

struct Options {
	// files
	std::string index_base, reads_file, out_file;
	// policy
	SimpleFunc score_min, n_ceil, ms_ival;
	int seed_len = 22, seed_mms = 0, n_seed_rounds = 2, max_dp_streak = 15;
	int khits = 1, mhits = 50;
	bool saw_k = false, all_hits = false, local = false;
	bool nofw = false, norc = false;
	bool sam_no_qname_trunc = false;
	bool qc_filter = false, ignore_quals = false, no_1mm_upfront = false, xeq = false, omit_sec_seq = false, phred64 = false;
	bool solexa_quals = false;           // --solexa-quals: 64-based Solexa (log-odds) quality characters (qual.h:105-123)
	int format = 0;               // 0 FASTQ, 1 FASTA (-f), 2 raw (-r), 3 -c, 4 tab5/6, 5 qseq, 6 FASTA-continuous (-F), 7 unaligned BAM (-b)
	bool preserve_tags = false;   // --preserve-tags: a BAM record's optional fields are printed after the aligner's (pat.cpp:1503, sam.cpp:881)
	bool align_paired_reads = false;   // --align-paired-reads: take the paired records of a BAM file instead of the unpaired ones (pat.cpp:1417-1427)
	bool sam_append_comment = false;   // --sam-append-comment: the FASTA/FASTQ comment (name after the first blank) closes the SAM record (sam.h:415)
	// effort knobs behind the presets (bt2_search.cpp:463-492, 1274-1310, 1461-1477); -k scales the limits further (to_params)
	int max_iters = 400, max_ug = 300, max_dp = 300, seed_boost_thresh = 300, tighten = 3;
	bool do_ungapped = true, do_extend = true;
	bool sc_unmapped = false;     // --soft-clipped-unmapped-tlen: TLEN without the soft-clipped ends (aligner_result.h:894-909)
	int fc_len = 0, fc_freq = 1;  // -F k:<len>,i:<freq>
	int trim5 = 0, trim3 = 0;
	int trim_to_side = 3, trim_to_len = -1;   // --trim-to [3:|5:]<len>
	int mp_max = 6, mp_min = 2, np = 1, rdg_const = 5, rdg_linear = 3, rfg_const = 5, rfg_linear = 3, gbar = 4, maxhalf = 15;
	int ma = 0;                   // match bonus (--ma; 2 in --local mode, always 0 end to end)
	bool set_D = false, set_R = false, set_L = false, set_i = false, set_score_min = false, set_ma = false;
	bool mm_const = false;        // MMP=Cxx: constant mismatch penalty (policy string / --bwa-sw-like)
	bool mm_rounded = false;      // MMP=R: Maq-style rounded quality (0/10/20/30)
	bool bwa_sw_like = false;     // --bwa-sw-like: minimum score = a*max(T, c*ln(len)) (bt2_search.cpp:3341-3350)
	bool report_overhangs = false;
	bool det_seeds = false;       // -d / --deterministic-seeds
	int seed_cache_mb = 20;       // --seed-cache-sz (bt2_search.cpp:457,1184)
	bool passthrough = false;     // --passthrough: every SAM line is followed by the read's original text (the Perl wrapper's --un/--al/--un-conc/--al-conc)
	bool no_exact_upfront = false;
	// paired-end input and policy (bt2_search.cpp:1185-1215; PairedEndPolicy pe.h:169)
	std::string mate1_file, mate2_file, interleaved_file;
	bool paired = false;
	bool mixed_unpaired = false;      // -U next to -1/-2 (or --interleaved): the pairs first, then the unpaired reads, one summary (pat.cpp:305-420)
	int min_insert = 0, max_insert = 500;
	bool mate1fw = true, mate2fw = false;         // --fr (default) / --rf / --ff
	bool no_mixed = false, no_discordant = false, dovetail = false, no_contain = false, no_overlap = false;
	std::string rg_id, rgs, rg_optflag;   // @RG header pieces and the per-record RG:Z: flag (bt2_search.cpp:1418-1436)
	uint32_t seed = 0;
	int threads = 1;
	bool reorder = false, timing = false, no_unal = false, quiet = false, sam_no_hd = false, sam_no_sq = false;
	uint64_t skip = 0, upto = std::numeric_limits<uint64_t>::max();
	std::string cmdline;      // for @PG
	std::string preset = "sensitive";

	Options() {
		score_min.init(2, (double)-0.6f, (double)-0.6f);
		n_ceil.init(2, (double)0.0f, std::numeric_limits<double>::max(), (double)0.0f, (double)0.15f);
		ms_ival.init(3, (double)1.0f, std::numeric_limits<double>::max(), (double)0.0f, (double)1.15f);
		resolve_preset();
	}
	bool apply_preset(const std::string& p) {
		// presets.cpp:33-91 (PresetsV0); the *-local flavours are selected by --local (the "%LOCAL%" substitution, bt2_search.cpp:1031)
		static const char* names[] = {"very-fast", "fast", "sensitive", "very-sensitive"};
		for (const char* n : names) if (p == n) { preset = p; return true; }
		for (const char* n : names) if (p == std::string(n) + "-local") { preset = n; local = true; return true; }
		return false;
	}
	// called once after all options are parsed
	void resolve_preset() {
		const std::string p = preset + (local ? "-local" : "");
		struct Row { const char* name; int dps, rounds, seedlen; double c, l; };
		static const Row rows[] = {
			{"very-fast", 5, 1, 22, 0.0, 2.50}, {"fast", 10, 2, 22, 0.0, 2.50}, {"sensitive", 15, 2, 22, 1.0, 1.15}, {"very-sensitive", 20, 3, 20, 1.0, 0.50},
			{"very-fast-local", 5, 1, 25, 1.0, 2.00}, {"fast-local", 10, 2, 22, 1.0, 1.75}, {"sensitive-local", 15, 2, 20, 1.0, 0.75}, {"very-sensitive-local", 20, 3, 20, 1.0, 0.50}};
		for (const Row& r : rows) if (p == r.name) {
			if (!set_D) max_dp_streak = r.dps;
			if (!set_R) n_seed_rounds = r.rounds;
			if (!set_L) seed_len = r.seedlen;
			if (!set_i) { ms_ival.type = 3; ms_ival.C = r.c; ms_ival.L = r.l; }
		}
		if (local) {
			if (!set_score_min) score_min.init(4, (double)20.0f, (double)8.0f);     // G,20,8 (DEFAULT_MIN_*_LOCAL)
			if (!set_ma) ma = 2;                                                    // DEFAULT_MATCH_BONUS_LOCAL
		} else ma = 0;
	}
	void to_params(AlignParams& P, bool large_index) const {
		// Scoring (scoring.h:60-170): type 3 = Phred-scaled mismatch penalty, anything else = constant mm_max;
		// gap of length n costs const + n * linear
		const bool cmm = ignore_quals || mm_const;
		P.mm_type = mm_rounded ? 2 : (cmm ? 1 : 3); P.mm_max = mp_max; P.mm_min = (cmm && !mm_rounded) ? mp_max : mp_min; P.n_pen = np;
		P.rdgapo = rdg_const + rdg_linear; P.rdgape = rdg_linear; P.rfgapo = rfg_const + rfg_linear; P.rfgape = rfg_linear;
		P.gapbar = gbar; P.match_bonus = local ? ma : 0;
		// (-a: every alignment found, up to what a result record of the many-alignments class holds)
		P.khits = all_hits ? BT2G_MAX_KHITS : khits; P.mhits = (saw_k || all_hits) ? 0 : mhits; P.all_hits = all_hits ? 1 : 0; P.seed_mms = seed_mms; P.overhang = report_overhangs ? 1 : 0;
		P.paired = paired ? 1 : 0;
		P.pe_policy = (mate1fw && mate2fw) ? 1 : ((!mate1fw && !mate2fw) ? 2 : (mate1fw ? 3 : 4));   // gMate1fw/gMate2fw -> PE_POLICY_* (bt2_search.cpp:1841-1851)
		P.pe_maxfrag = max_insert; P.pe_minfrag = min_insert;
		P.pe_flags = (dovetail ? BT2G_PE_DOVETAIL_OK : 0) | (no_contain ? 0 : BT2G_PE_CONTAIN_OK) | (no_overlap ? 0 : BT2G_PE_OLAP_OK) | BT2G_PE_EXPAND |
		             (no_discordant ? 0 : BT2G_PE_DISCORD) | (no_mixed ? 0 : BT2G_PE_MIXED) | (mate1fw ? BT2G_PE_MATE1FW : 0) | (mate2fw ? BT2G_PE_MATE2FW : 0);
		P.max_mate_streak = 10;
		P.det_seeds = det_seeds ? 1 : 0;
		P.seed_cache_mb = seed_cache_mb;
		P.profile = 0;      // the drop-in binary never asks for the worker's phase timers
		P.max_seeds = 0;    // set per batch by the driver (bt2g_search.cpp)
		P.max_dp_cols = 0;  // likewise (paired batches: the widest opposite-mate window)
		P.max_dp_streak = max_dp_streak; P.max_ug = max_ug; P.max_dp = max_dp; P.max_iters = max_iters;
		if (all_hits) {
			// -a lifts every effort limit (bt2_search.cpp:3457-3463)
			P.max_dp_streak = P.max_ug = P.max_dp = P.max_iters = P.max_mate_streak = 0x7fffffff;
		} else if (khits > 1) {
			// streak/limit scaling with -k (bt2_search.cpp:3452-3476): maxStreakIncr=10, maxItersIncr=20
			P.max_dp_streak += (khits - 1) * 10; P.max_mate_streak += (khits - 1) * 10;
			P.max_ug += (khits - 1) * 20; P.max_dp += (khits - 1) * 20; P.max_iters += (khits - 1) * 20;
		}
		P.n_seed_rounds = n_seed_rounds; P.seed_boost_thresh = seed_boost_thresh; P.tighten = tighten; P.maxhalf = maxhalf;
		P.nofw = nofw; P.norc = norc;
		P.do_exact_upfront = no_exact_upfront ? 0 : 1; P.do_1mm_upfront = no_1mm_upfront ? 0 : 1; P.do_ungapped = do_ungapped ? 1 : 0;
		// bit 0: extend seed hits; bit 1: left only -- the reference loads the mirror index only for -N > 0 or the 1-mm
		// up-front search (bt2_search.cpp:4841) and SwDriver::extend skips the right extension without it (:403)
		P.do_extend = do_extend ? (1 | ((seed_mms == 0 && no_1mm_upfront) ? 2 : 0)) : 0;
		P.large_index = large_index ? 1 : 0;
	}
};

// Non-owning view of characters that live in a batch-level arena (no per-read allocations on the hot host path)
struct StrView {
	const char* p = nullptr;
	uint32_t n = 0;
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	char operator[](size_t i) const { return p[i]; }
	const char* data() const { return p; }
	const char* begin() const { return p; }
	const char* end() const { return p + n; }
	std::string str() const { return std::string(p, n); }
	void set(const char* p_, size_t n_) { p = p_; n = (uint32_t)n_; }
};

struct ReadRec {
	StrView name;
	StrView seq;        // codes 0..4
	StrView qual;       // ASCII phred+33
	StrView orig;       // --passthrough: the record's original text (Read::readOrigBuf)
	StrView tags;       // --preserve-tags: the optional fields of the BAM record, in BAM's binary form (Read::preservedOptFlags)
	char filter = '1';  // QSEQ filter field ('0' = failed the instrument's QC; --qc-filter)
};

// asc2dna (alphabet.cpp:142): A/C/G/T (either case) -> 0..3, every other letter -> 4
inline int asc2code(int c) {
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// genRandSeed (pat.cpp:45-84)
inline uint32_t gen_rand_seed(const ReadRec& r, uint32_t seed) {
	uint32_t rseed = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	const size_t qlen = r.seq.size();
	for (size_t i = 0; i < qlen; i++) rseed ^= ((uint32_t)(int)r.seq[i] << ((i & 15) << 1));
	for (size_t i = 0; i < qlen; i++) rseed ^= (uint32_t)((int)r.qual[i] << ((i & 3) << 3));
	for (size_t i = 0; i < r.name.size(); i++) {
		const int p = (int)r.name[i];
		if (p == '/') break;
		rseed ^= (uint32_t)(p << ((i & 3) << 3));
	}
	return rseed;
}

// per-read derived parameters (bt2_search.cpp:3352-3450)
// the part of a read's parameters that depends on its length only (the reader computes it once per length, not once per read)
struct LenParams { int64_t minsc; size_t maxns; int nceil, interval; bool scfilt, lenfilt; };
inline LenParams compute_len_params(const Options& o, size_t len) {
	LenParams q;
	int64_t minsc;
	if (o.bwa_sw_like) {
		// "a*max{T,c*log(l)}" in float, as the reference evaluates it (bt2_search.cpp:3341-3350; T = 30, c = 5.5)
		const float a = (float)o.ma, T = 30.0f, c = 5.5f;
		const float v1 = a * T, v2 = (float)((double)(a * c) * std::log((double)len));
		minsc = (int64_t)std::max(v1, v2);
	} else {
		minsc = o.score_min.f<int64_t>((double)len);
		if (o.local) { if (minsc < 0) minsc = 0; } else if (minsc > 0) minsc = 0;          // bt2_search.cpp:3352-3372
	}
	q.minsc = minsc;
	q.maxns = o.n_ceil.f<size_t>((double)len);      // N filter (Scoring::nFilter)
	q.scfilt = (int64_t)len * (o.local ? o.ma : 0) >= minsc;       // Scoring::scoreFilter: perfect score must reach minsc
	q.lenfilt = !(len <= (size_t)o.seed_mms || len < 2);
	int nceil = o.n_ceil.f<int>((double)len);
	if (nceil > (int)len) nceil = (int)len;
	q.nceil = nceil;
	int interval = o.ms_ival.f<int>((double)len);
	if (interval < 1) interval = 1;
	q.interval = interval;
	return q;
}
inline ReadParams compute_read_params(const Options& o, const ReadRec& r, const LenParams& q) {
	ReadParams p;
	const size_t len = r.seq.size();
	p.minsc = (int32_t)q.minsc;
	size_t ns = 0;
	bool nfilt = true;
	for (size_t i = 0; i < len; i++) if (r.seq[i] == 4) { ns++; if (ns > q.maxns) { nfilt = false; break; } }
	const bool qcfilt = !(o.qc_filter && r.filter == '0');
	p.filt = (nfilt ? 1u : 0u) | (q.scfilt ? 2u : 0u) | (q.lenfilt ? 4u : 0u) | (qcfilt ? 8u : 0u);
	p.nceil = q.nceil;
	p.interval = q.interval;
	p.seedlen = o.seed_len;
	p.seed = gen_rand_seed(r, o.seed);
	return p;
}
inline ReadParams compute_read_params(const Options& o, const ReadRec& r) { return compute_read_params(o, r, compute_len_params(o, r.seq.size())); }

// ---------------------------------------------------------------------------------------
// SAM
struct RefInfo { std::vector<std::string> names; std::vector<uint64_t> lens; };

template <typename Str>
inline void sam_print_name(std::string& o, const Str& name, bool truncate) {
	size_t n = name.size();
	if (truncate && n > 255) n = 255;
	for (size_t i = 0; i < n; i++) { if (truncate && isspace((unsigned char)name[i])) break; o.push_back(name[i]); }
}

inline void sam_header(std::string& o, const RefInfo& ref, const std::string& cmdline, bool hd = true, bool sq = true,
                       const std::string& rg_id = "", const std::string& rgs = "") {
	if (hd) o += "@HD\tVN:1.5\tSO:unsorted\tGO:query\n";
	if (sq) for (size_t i = 0; i < ref.names.size(); i++) {
		o += "@SQ\tSN:";
		sam_print_name(o, ref.names[i], true);
		o += "\tLN:" + std::to_string(ref.lens[i]) + "\n";
	}
	if (!rg_id.empty()) o += "@RG" + rg_id + rgs + "\n";       // SamConfig::printHeader (sam.cpp)
	o += "@PG\tID:bowtie2\tPN:bowtie2\tVN:2.5.5\tCL:\"" + cmdline + "\"\n";
}

// BowtieMapq2::mapq for an unpaired, primary, end-to-end alignment (unique.h:185-330)
inline int mapq_v2(const Options& o, size_t rdlen, int64_t best, bool has_secbest, int64_t secbest_in, bool pair = false, size_t ordlen = 0) {
	// a concordant / discordant pair is rated on the sum of both mates' scores (unique.h:207-215)
	int64_t scPer = o.local ? (int64_t)rdlen * o.ma : 0;
	if (pair && o.local) scPer += (int64_t)ordlen * o.ma;
	if (o.local) {
		// non-monotone branch (unique.h:333-383)
		int64_t scMin = o.score_min.f<int64_t>((double)(float)rdlen);
		if (pair) scMin += o.score_min.f<int64_t>((double)(float)ordlen);
		const int64_t diff = std::max<int64_t>(1, scPer - scMin);
		const int64_t bestOver = best - scMin;
		if (!has_secbest) {
			if      (bestOver >= diff * (double)0.8f) return 44;
			else if (bestOver >= diff * (double)0.7f) return 42;
			else if (bestOver >= diff * (double)0.6f) return 41;
			else if (bestOver >= diff * (double)0.5f) return 36;
			else if (bestOver >= diff * (double)0.4f) return 28;
			else if (bestOver >= diff * (double)0.3f) return 24;
			return 22;
		}
		const int64_t bestdiff = std::llabs(std::llabs(best) - std::llabs(secbest_in));
		if      (bestdiff >= diff * (double)0.9f) return 40;
		else if (bestdiff >= diff * (double)0.8f) return 39;
		else if (bestdiff >= diff * (double)0.7f) return 38;
		else if (bestdiff >= diff * (double)0.6f) return 37;
		else if (bestdiff >= diff * (double)0.5f) return bestOver == diff ? 35 : (bestOver >= diff * (double)0.50f ? 25 : 20);
		else if (bestdiff >= diff * (double)0.4f) return bestOver == diff ? 34 : (bestOver >= diff * (double)0.50f ? 21 : 19);
		else if (bestdiff >= diff * (double)0.3f) return bestOver == diff ? 33 : (bestOver >= diff * (double)0.5f ? 18 : 16);
		else if (bestdiff >= diff * (double)0.2f) return bestOver == diff ? 32 : (bestOver >= diff * (double)0.5f ? 17 : 12);
		else if (bestdiff >= diff * (double)0.1f) return bestOver == diff ? 31 : (bestOver >= diff * (double)0.5f ? 14 : 9);
		else if (bestdiff > 0) return bestOver >= diff * (double)0.5f ? 11 : 2;
		return bestOver >= diff * (double)0.5f ? 1 : 0;
	}
	int64_t scMin = o.score_min.f<int64_t>((double)(float)rdlen);
	if (pair) scMin += o.score_min.f<int64_t>((double)(float)ordlen);
	int64_t secbest = scMin - 1;
	const int64_t diff = std::max<int64_t>(1, scPer - scMin);
	int ret = 0;
	const int64_t bestOver = best - scMin;
	if (!has_secbest) {
		if      (bestOver >= diff * (double)0.8f) ret = 42;
		else if (bestOver >= diff * (double)0.7f) ret = 40;
		else if (bestOver >= diff * (double)0.6f) ret = 24;
		else if (bestOver >= diff * (double)0.5f) ret = 23;
		else if (bestOver >= diff * (double)0.4f) ret = 8;
		else if (bestOver >= diff * (double)0.3f) ret = 3;
		else ret = 0;
	} else {
		secbest = secbest_in;
		const int64_t bestdiff = std::abs(std::abs(best) - std::abs(secbest));
		if (bestdiff >= diff * (double)0.9f) ret = (bestOver == diff) ? 39 : 33;
		else if (bestdiff >= diff * (double)0.8f) ret = (bestOver == diff) ? 38 : 27;
		else if (bestdiff >= diff * (double)0.7f) ret = (bestOver == diff) ? 37 : 26;
		else if (bestdiff >= diff * (double)0.6f) ret = (bestOver == diff) ? 36 : 22;
		else if (bestdiff >= diff * (double)0.5f) {
			if (bestOver == diff) ret = 35; else if (bestOver >= diff * (double)0.84f) ret = 25;
			else if (bestOver >= diff * (double)0.68f) ret = 16; else ret = 5;
		} else if (bestdiff >= diff * (double)0.4f) {
			if (bestOver == diff) ret = 34; else if (bestOver >= diff * (double)0.84f) ret = 21;
			else if (bestOver >= diff * (double)0.68f) ret = 14; else ret = 4;
		} else if (bestdiff >= diff * (double)0.3f) {
			if (bestOver == diff) ret = 32; else if (bestOver >= diff * (double)0.88f) ret = 18;
			else if (bestOver >= diff * (double)0.67f) ret = 15; else ret = 3;
		} else if (bestdiff >= diff * (double)0.2f) {
			if (bestOver == diff) ret = 31; else if (bestOver >= diff * (double)0.88f) ret = 17;
			else if (bestOver >= diff * (double)0.67f) ret = 11; else ret = 0;
		} else if (bestdiff >= diff * (double)0.1f) {
			if (bestOver == diff) ret = 30; else if (bestOver >= diff * (double)0.88f) ret = 12;
			else if (bestOver >= diff * (double)0.67f) ret = 7; else ret = 0;
		} else if (bestdiff > 0) {
			ret = (bestOver >= diff * (double)0.67f) ? 6 : 2;
		} else {
			ret = (bestOver >= diff * (double)0.67f) ? 1 : 0;
		}
	}
	return ret;
}

// One SAM record for an unpaired read (AlnSinkSam::appendMate + printAlignedOptFlags)
// decimal append without the std::to_string temporaries
inline void app_int(std::string& o, int64_t v) {
	char buf[24];
	int n = 0;
	uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
	do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) o.push_back('-');
	while (n) o.push_back(buf[--n]);
}

// What AlnSinkSam::appendMate needs to know about the other mate (aln_sink.cpp:1889-2124; AlnFlags aligner_result.h:1580)
struct MateOut {
	bool mate1 = true;
	int kind = 0;                       // 0 mate reported on its own (YT:Z:UP), 1 concordant (CP), 2 discordant (DP)
	bool opp_aligned = false, opp_fw = true;
	const AlnRes* rso = nullptr;        // the other mate's alignment that goes with this line
	int64_t orefid = -1, orefoff = -1;  // unaligned mate: where the other mate went
	size_t ordlen = 0;
	int64_t pair_best = 0, pair_secbest = 0;
	bool pair_has_secbest = false;
};

// TLEN (AlnRes::setFragmentLength, aligner_result.h:1311-1345): extents include soft-trimmed bases
inline int64_t sam_fragment_length(const AlnRes& a, const AlnRes& b, bool a_is_mate1, bool sc_unmapped = false) {
	auto ext = [sc_unmapped](const AlnRes& r, int64_t& st, int64_t& en) {
		st = r.refoff; en = r.refoff + (int64_t)r.rfextent - 1;
		if (sc_unmapped) return;          // --soft-clipped-unmapped-tlen (getExtendedCoords, aligner_result.h:900)
		st -= r.fw ? r.trim5p : r.trim3p;
		en += r.fw ? r.trim3p : r.trim5p;
	};
	int64_t st, en, ost, oen;
	ext(a, st, en); ext(b, ost, oen);
	bool up;
	if (st == ost) up = (a.fw && b.fw && a_is_mate1) || (a.fw && !b.fw);
	else up = st < ost;
	const int64_t lo = std::min(st, ost), hi = std::max(en, oen);
	const int64_t fl = 1 + hi - lo;
	return up ? fl : -fl;
}


// --preserve-tags: the optional fields of a BAM record as SAM text (SamConfig::printPreservedOptFlags, sam.cpp:881-955).
// Integer types all print as "i"; a B array prints its element type and the comma-separated values; anything else ends the field list
// the way the reference's switch falls through (no value, next tag read from the following byte).
inline void sam_print_bam_tags(std::string& o, const StrView& t) {
	const char* b = t.data();
	size_t i = 0;
	const size_t len = t.size();
	uint32_t count = 1;                  // a B array's length stays in force for the fields after it, as in the reference
	auto put = [&](auto proto) {
		typedef decltype(proto) T;
		for (uint32_t k = 0; k < (count ? count : 1u); k++) {
			if (i + (k + 1) * sizeof(T) > len) break;          // (the reference reads on past the record here)
			T v; memcpy(&v, b + i + k * sizeof(T), sizeof(T));
			o += std::to_string(v);
			if (k + 1 < count) o.push_back(',');
		}
		i += sizeof(T) * count;
	};
	while (i + 3 <= len) {
		o.push_back('\t');
		o.append(b + i, 2); i += 2;
		char ty = b[i];
		if (ty == 'B') {
			if (i + 6 > len) break;              // array header cut short: malformed record, stop here
			ty = b[i + 1]; i += 2;
			memcpy(&count, b + i, 4); i += 4;
			o += ":B:"; o.push_back(ty); o.push_back(',');
		} else {
			o.push_back(':');
			o.push_back((ty == 'c' || ty == 'C' || ty == 'i' || ty == 'I' || ty == 's' || ty == 'S') ? 'i' : ty);
			i += 1;
			o.push_back(':');
		}
		switch (ty) {
			case 'A': put((char)0); break;
			case 'c': put((int8_t)0); break;
			case 'C': put((uint8_t)0); break;
			case 's': put((int16_t)0); break;
			case 'S': put((uint16_t)0); break;
			case 'i': put((int32_t)0); break;
			case 'I': put((uint32_t)0); break;
			case 'f': put((float)0); break;
			case 'Z': while (i < len && b[i]) o.push_back(b[i++]); i++; break;
			default: break;
		}
	}
}

// --sam-append-comment (SamConfig::printComment / isIllumina, sam.h:415-463): what follows the first blank of the read name closes the
// record; a CASAVA 1.8 comment "<1|2>:<Y|N>:<even number>:<barcode>" gets the BC:Z: tag name, anything else is taken as written
inline void sam_print_comment(std::string& o, const StrView& name) {
	size_t i = 0;
	while (i < name.size() && !isspace((unsigned char)name[i])) i++;
	o.push_back('\t');
	if (i >= name.size()) return;
	const std::string c(name.data() + i + 1, name.size() - i - 1);
	bool illumina = true;
	{
		int field = 0;
		size_t start = 0;
		for (size_t e = 0; e < c.size() && c[e] != ' ' && illumina; e++) {
			if (c[e] != ':') continue;
			const std::string f = c.substr(start, e - start);
			char* endp = nullptr;
			if (field == 0) { const long v = strtol(f.c_str(), &endp, 10); if (*endp || (v != 1 && v != 2)) illumina = false; }
			else if (field == 1) { if (c[start] != 'N' && c[start] != 'Y') illumina = false; }
			else if (field == 2) { const long v = strtol(f.c_str(), &endp, 10); if (*endp || v % 2 != 0) illumina = false; }
			else illumina = false;
			start = e + 1; field++;
		}
	}
	if (illumina) o += "BC:Z:";
	o += c;
}

inline void sam_record(std::string& o, const Options& opt, const RefInfo& ref, const ReadRec& rd,
                       const ReadResult& rr, const AlnRes* aln, bool primary, const MateOut* mo = nullptr) {
	static const char* DNA = "ACGTN";
	const size_t len = rd.seq.size();
	sam_print_name(o, rd.name, !opt.sam_no_qname_trunc);
	if (mo) {   // printReadName(omitSlashMate): a trailing /1, /2 or /3 is dropped for mates (sam.h:329-336)
		const size_t ol = o.size();
		if (ol >= 2 && o[ol - 2] == '/' && (o[ol - 1] == '1' || o[ol - 1] == '2' || o[ol - 1] == '3')) o.resize(ol - 2);
	}
	o.push_back('\t');
	int fl = 0;
	if (mo) {
		fl |= 1;
		if (mo->kind == 1) fl |= 2;
		if (!mo->opp_aligned) fl |= 8;
		fl |= mo->mate1 ? 64 : 128;
		if (mo->opp_aligned) { const bool ofw = mo->rso ? mo->rso->fw != 0 : mo->opp_fw; if (!ofw) fl |= 32; }
	}
	if (!primary) fl |= 256;
	if (aln && !aln->fw) fl |= 16;
	if (!aln) fl |= 4;
	app_int(o, (int64_t)(fl)); o.push_back('\t');
	if (aln) { sam_print_name(o, ref.names[aln->refid], true); o.push_back('\t'); app_int(o, (int64_t)(aln->refoff + 1)); o.push_back('\t'); }
	else if (mo && mo->orefid != -1) { sam_print_name(o, ref.names[mo->orefid], true); o.push_back('\t'); app_int(o, mo->orefoff + 1); o.push_back('\t'); }
	else o += "*\t0\t";
	// stacked alignment (StackedAln::init / leftAlign / buildCigar / buildMdz)
	static thread_local std::string stRef, stRel, stRead, md;
	static thread_local std::vector<Edit> ed;
	stRef.clear(); stRel.clear(); stRead.clear();
	// An alignment without gaps (nearly all of them) needs no stacked alignment: its CIGAR and MD:Z follow from the mismatch positions.
	bool nogap = aln != nullptr;
	if (aln) for (uint32_t i = 0; i < aln->nned; i++) if (aln->ned[i].type != EDIT_MM) { nogap = false; break; }
	if (aln) {
		// edits w.r.t. the upstream end: invert for rc (AlnRes::initStacked)
		ed.assign(aln->ned, aln->ned + aln->nned);
		size_t trimLS = aln->trim5p, trimRS = aln->trim3p;
		const size_t len_trimmed = len - trimLS - trimRS;
		if (!aln->fw) {
			for (size_t i = 0; i < ed.size() / 2; i++) std::swap(ed[i], ed[ed.size() - 1 - i]);
			for (auto& e : ed) e.pos = (uint16_t)(len_trimmed - e.pos - (e.type == EDIT_READ_GAP ? 0 : 1));
			std::swap(trimLS, trimRS);
		}
		auto s = [&](size_t i) -> int { return aln->fw ? (int)rd.seq[i] : comp4((int)rd.seq[len - 1 - i]); };
		size_t rdoff = trimLS;
		if (!nogap) for (const auto& e : ed) {
			const size_t pos = e.pos + trimLS;
			while (rdoff < pos) { const int c = s(rdoff++); stRef.push_back(DNA[c]); stRel.push_back('='); stRead.push_back(DNA[c]); }
			if (e.type == EDIT_MM) { const int c = s(rdoff++); stRef.push_back((char)e.chr); stRel.push_back('X'); stRead.push_back(DNA[c]); }
			else if (e.type == EDIT_REF_GAP) { const int c = s(rdoff++); stRef.push_back('-'); stRel.push_back('I'); stRead.push_back(DNA[c]); }
			else { stRef.push_back((char)e.chr); stRel.push_back('D'); stRead.push_back('-'); }
		}
		if (!nogap) while (rdoff < len - trimRS) { const int c = s(rdoff++); stRef.push_back(DNA[c]); stRel.push_back('='); stRead.push_back(DNA[c]); }
		// leftAlign(false)
		const size_t ln = stRef.size();
		for (size_t i = 0; i < ln; i++) {
			const char rel = stRel[i];
			if (rel != '=' && rel != 'X') {
				size_t glen = 1;
				for (size_t j = i + 1; j < ln; j++) { if (rel != stRel[j]) break; glen++; }
				size_t l = i - 1, r = l + glen;
				std::string& gp = (rel == 'I') ? stRef : stRead;
				const std::string& ngp = (rel == 'I') ? stRead : stRef;
				while (l > 0 && l < ln && ngp[l] == ngp[r]) {
					if (stRel[l] == 'X') break;
					std::swap(gp[l], gp[r]);
					std::swap(stRel[l], stRel[r]);
					l--; r--;
				}
				i += (glen - 1);
			}
		}
		// unique.h:199-205: a secondary line, or "one alignment found without looking for a second" (-k: canMax is
		// false, and the reference never sets `exhausted`), reports 255
		const bool can_max = !(opt.saw_k || opt.all_hits) && opt.mhits > 0;
		const bool summ_paired = mo && mo->kind != 0;
		const bool has_sec = summ_paired ? mo->pair_has_secbest : rr.has_secbest != 0;
		if (!primary || (!can_max && !has_sec)) o += "255";
		else if (summ_paired) app_int(o, (int64_t)mapq_v2(opt, len, mo->pair_best, has_sec, mo->pair_secbest, true, mo->ordlen));
		else app_int(o, (int64_t)(mapq_v2(opt, len, rr.best, rr.has_secbest != 0, rr.secbest)));
		o.push_back('\t');
		// CIGAR
		if (trimLS > 0) { app_int(o, (int64_t)(trimLS)); o.push_back('S'); }
		if (nogap) {
			if (!opt.xeq) { if (len_trimmed > 0) { app_int(o, (int64_t)len_trimmed); o.push_back('M'); } }
			else {
				size_t at = 0;
				for (size_t i = 0; i < ed.size();) {
					const size_t p0 = ed[i].pos;
					if (p0 > at) { app_int(o, (int64_t)(p0 - at)); o.push_back('='); }
					size_t run = 1;
					while (i + run < ed.size() && (size_t)ed[i + run].pos == p0 + run) run++;
					app_int(o, (int64_t)run); o.push_back('X');
					at = p0 + run; i += run;
				}
				if (len_trimmed > at) { app_int(o, (int64_t)(len_trimmed - at)); o.push_back('='); }
			}
			// MD:Z (buildMdz + writeMdz) for the no-gap case: match runs and mismatched reference characters, a 0 between adjacent
			// mismatches, before a leading and after a trailing one
			md.clear();
			size_t at = 0;
			bool mm_last = false, first_print = true;
			for (const auto& e : ed) {
				if ((size_t)e.pos > at) { app_int(md, (int64_t)((size_t)e.pos - at)); first_print = false; mm_last = false; }
				if (mm_last || first_print) md.push_back('0');
				md.push_back((char)e.chr);
				first_print = false; mm_last = true;
				at = (size_t)e.pos + 1;
			}
			if (len_trimmed > at) app_int(md, (int64_t)(len_trimmed - at));
			else if (mm_last) md.push_back('0');
		} else
		for (size_t i = 0; i < ln; i++) {
			char op = stRel[i];
			if (!opt.xeq && (op == 'X' || op == '=')) op = 'M';
			size_t run = 1;
			for (; i + run < ln; run++) { char op2 = stRel[i + run]; if (!opt.xeq && (op2 == 'X' || op2 == '=')) op2 = 'M'; if (op2 != op) break; }
			i += (run - 1);
			app_int(o, (int64_t)(run)); o.push_back(op);
		}
		if (trimRS > 0) { app_int(o, (int64_t)(trimRS)); o.push_back('S'); }
		o.push_back('\t');
	} else {
		o += "0\t*\t";
	}
	// RNEXT / PNEXT / TLEN
	if (aln && mo) {
		if (mo->rso && aln->refid != mo->rso->refid) { sam_print_name(o, ref.names[mo->rso->refid], true); o.push_back('\t'); } else o += "=\t";
		app_int(o, (mo->rso ? mo->rso->refoff : aln->refoff) + 1); o.push_back('\t');
		// fragment length only for pairs (setMateParams with the other mate), on the same reference or concordant
		if (mo->kind != 0 && mo->rso && (aln->refid == mo->rso->refid || mo->kind == 1)) app_int(o, sam_fragment_length(*aln, *mo->rso, mo->mate1, opt.sc_unmapped));
		else o.push_back('0');
		o.push_back('\t');
	} else if (mo && mo->orefid != -1) { o += "=\t"; app_int(o, mo->orefoff + 1); o += "\t0\t"; }
	else o += "*\t0\t0\t";
	// SEQ / QUAL
	if (len == 0) o.push_back('*');
	else if (!primary && opt.omit_sec_seq) o.push_back('*');
	else {
		const size_t o0 = o.size();
		o.resize(o0 + len);
		char* w = &o[o0];
		if (!aln || aln->fw) for (size_t i = 0; i < len; i++) w[i] = DNA[(int)rd.seq[i]];
		else for (size_t i = 0; i < len; i++) w[i] = DNA[comp4((int)rd.seq[len - 1 - i])];
	}
	o.push_back('\t');
	if (len == 0) o.push_back('*');
	else if (!primary && opt.omit_sec_seq) o.push_back('*');
	else if (!aln || aln->fw) o.append(rd.qual.data(), rd.qual.size());
	else {
		const size_t o0 = o.size();
		o.resize(o0 + len);
		char* w = &o[o0];
		for (size_t i = 0; i < len; i++) w[i] = rd.qual[len - 1 - i];
	}
	o.push_back('\t');
	// optional fields
	if (aln) {
		o += "AS:i:"; app_int(o, aln->score);
		// XS:i: for mates the best alignment of this mate at another locus than the chosen pair's; a mate reported
		// on its own (UP) never carries it -- the reference looks up the pair fields, which only pairs fill (sam.cpp:144-150)
		if (rr.has_secbest && (!mo || mo->kind != 0)) { o += "\tXS:i:"; app_int(o, rr.secbest); }
		o += "\tXN:i:"; app_int(o, aln->refns);
		size_t num_mm = 0, num_go = 0, num_gx = 0;
		for (size_t i = 0; i < aln->nned; i++) {
			const Edit& e = aln->ned[i];
			if (e.type == EDIT_MM) num_mm++;
			else if (e.type == EDIT_READ_GAP) {
				num_go++; num_gx++;
				while (i + 1 < aln->nned && aln->ned[i + 1].pos == aln->ned[i].pos && aln->ned[i + 1].type == EDIT_READ_GAP) { i++; num_gx++; }
			} else if (e.type == EDIT_REF_GAP) {
				num_go++; num_gx++;
				while (i + 1 < aln->nned && aln->ned[i + 1].pos == aln->ned[i].pos + 1 && aln->ned[i + 1].type == EDIT_REF_GAP) { i++; num_gx++; }
			}
		}
		o += "\tXM:i:"; app_int(o, (int64_t)num_mm); o += "\tXO:i:"; app_int(o, (int64_t)num_go); o += "\tXG:i:"; app_int(o, (int64_t)num_gx);
		o += "\tNM:i:"; app_int(o, aln->nned);
		// MD:Z (buildMdz + writeMdz); the reference prints it before YT:Z
		if (!nogap) {
			md.clear();
			bool mm_last = false, rdgap_last = false, first_print = true;
			const size_t ln = stRef.size();
			for (size_t i = 0; i < ln; i++) {
				const char op = stRel[i];
				if (op == '=') {
					size_t run = 1, nins = 0;
					for (; i + run < ln; run++) {
						if (stRel[i + run] == '=') {} else if (stRel[i + run] == 'I') nins++; else break;
					}
					i += (run - 1);
					if (run - nins > 0) { app_int(md, (int64_t)(run - nins)); first_print = false; mm_last = false; rdgap_last = false; }
				} else if (op == 'X') {
					if (rdgap_last || mm_last || first_print) md.push_back('0');
					md.push_back(stRef[i]);
					first_print = false; mm_last = true; rdgap_last = false;
				} else if (op == 'D') {
					if (mm_last || first_print) md.push_back('0');
					if (!rdgap_last) md.push_back('^');
					md.push_back(stRef[i]);
					first_print = false; mm_last = false; rdgap_last = true;
				}
			}
			if (mm_last || rdgap_last) md.push_back('0');
		}
		o += "\tMD:Z:"; o += md;
		if (mo && mo->kind != 0 && mo->rso) { o += "\tYS:i:"; app_int(o, mo->rso->score); }
		// YF would go here for filtered reads, but filtered reads never align
		o += mo ? (mo->kind == 1 ? "\tYT:Z:CP" : (mo->kind == 2 ? "\tYT:Z:DP" : "\tYT:Z:UP")) : "\tYT:Z:UU";
	} else {
		o += mo ? "YT:Z:UP" : "YT:Z:UU";
		const uint32_t f = rr.filt;
		const char* flag = "";
		if (!(f & 4)) flag = "LN"; else if (!(f & 1)) flag = "NS"; else if (!(f & 2)) flag = "SC"; else if (!(f & 8)) flag = "QC";
		if (flag[0]) { o += "\tYF:Z:"; o += flag; }
	}
	if (!opt.rg_optflag.empty()) { o.push_back('\t'); o += opt.rg_optflag; }
	if (!rd.tags.empty()) sam_print_bam_tags(o, rd.tags);
	if (opt.sam_append_comment) sam_print_comment(o, rd.name);
	o.push_back('\n');
	if (opt.passthrough) {      // samc_.passthrough() (aln_sink.cpp:2118): newlines and '%' percent-encoded (sam.h:290)
		for (size_t i = 0; i < rd.orig.size(); i++) {
			const unsigned char ch = (unsigned char)rd.orig[i];
			if (ch == 10 || ch == 13 || ch == '%') { o.push_back('%'); o.push_back("0123456789ABCDEF"[ch >> 4]); o.push_back("0123456789ABCDEF"[ch & 15]); }
			else o.push_back((char)ch);
		}
		o.push_back('\n');
	}
}

// SAM lines of one pair, in the order AlnSinkWrap::finishRead / AlnSink::reportHits emit them (aln_sink.cpp:700-1390,
// aln_sink.h:646-728).  a1 / a2: the nreport alignments of each mate's record.
inline void sam_pair_records(std::string& o, const Options& opt, const RefInfo& ref, const ReadRec& rd1, const ReadRec& rd2,
                             const ReadResult& rr1, const ReadResult& rr2, const AlnRes* const* a1, const AlnRes* const* a2) {
	const ReadRec* rd[2] = {&rd1, &rd2};
	const ReadResult* rr[2] = {&rr1, &rr2};
	const AlnRes* const* al[2] = {a1, a2};
	if (rr1.pair_type != 0) {
		// concordant or discordant: alignment i of mate 1 goes with alignment i of mate 2
		for (uint32_t i = 0; i < rr1.nreport; i++) {
			for (int m = 0; m < 2; m++) {
				MateOut mo;
				mo.mate1 = m == 0; mo.kind = rr1.pair_type; mo.opp_aligned = true; mo.rso = al[m ^ 1][i]; mo.opp_fw = mo.rso->fw != 0;
				mo.ordlen = rd[m ^ 1]->seq.size();
				mo.pair_best = rr[m]->pair_best; mo.pair_secbest = rr[m]->pair_secbest; mo.pair_has_secbest = (rr[m]->pair_flags & 2) != 0;
				sam_record(o, opt, ref, *rd[m], *rr[m], al[m][i], i == 0, &mo);
			}
		}
		return;
	}
	// each mate on its own: all of mate 1's lines, then mate 2's; an unaligned mate borrows the other's primary position
	const AlnRes* pri[2] = {rr1.aligned ? a1[0] : nullptr, rr2.aligned ? a2[0] : nullptr};
	int64_t orefid = -1, orefoff = -1;
	for (int m = 0; m < 2; m++) {
		if (!rr[m]->aligned) continue;
		for (uint32_t i = 0; i < rr[m]->nreport; i++) {
			MateOut mo;
			mo.mate1 = m == 0; mo.kind = 0; mo.opp_aligned = pri[m ^ 1] != nullptr; mo.rso = pri[m ^ 1]; mo.opp_fw = !pri[m ^ 1] || pri[m ^ 1]->fw != 0;
			mo.ordlen = rd[m ^ 1]->seq.size();
			sam_record(o, opt, ref, *rd[m], *rr[m], al[m][i], i == 0, &mo);
		}
		orefid = al[m][0]->refid; orefoff = al[m][0]->refoff;
	}
	for (int m = 0; m < 2; m++) {
		if (rr[m]->aligned || opt.no_unal) continue;
		MateOut mo;
		mo.mate1 = m == 0; mo.kind = 0; mo.opp_aligned = pri[m ^ 1] != nullptr; mo.opp_fw = pri[m ^ 1] ? pri[m ^ 1]->fw != 0 : false;
		if (rr[m ^ 1]->aligned) { mo.orefid = orefid; mo.orefoff = orefoff; }
		mo.ordlen = rd[m ^ 1]->seq.size();
		sam_record(o, opt, ref, *rd[m], *rr[m], nullptr, true, &mo);
	}
}

struct AlnSummary {
	uint64_t nread = 0, n0 = 0, nuni = 0, nrep = 0;
	// aln_sink.cpp:985-1013,486-512: "exactly 1 time" = one alignment found and not over the -M ceiling
	void merge(const AlnSummary& o) { nread += o.nread; n0 += o.n0; nuni += o.nuni; nrep += o.nrep; }
	void add(const ReadResult& r) { nread++; if (!r.aligned) n0++; else if (r.maxed || r.nalns > 1) nrep++; else nuni++; }
	void print(FILE* f) const {
		auto pct = [](uint64_t a, uint64_t b) { char buf[32]; snprintf(buf, sizeof buf, "%.2f%%", b ? 100.0 * (double)a / (double)b : 0.0); return std::string(buf); };
		if (nread == 0) fprintf(f, "0 reads\n");        // no "of these" and no section for an empty run (aln_sink.cpp:364-370, 457)
		else {
			fprintf(f, "%llu reads; of these:\n", (unsigned long long)nread);
			fprintf(f, "  %llu (%s) were unpaired; of these:\n", (unsigned long long)nread, pct(nread, nread).c_str());
			fprintf(f, "    %llu (%s) aligned 0 times\n", (unsigned long long)n0, pct(n0, nread).c_str());
			fprintf(f, "    %llu (%s) aligned exactly 1 time\n", (unsigned long long)nuni, pct(nuni, nread).c_str());
			fprintf(f, "    %llu (%s) aligned >1 times\n", (unsigned long long)nrep, pct(nrep, nread).c_str());
		}
		fprintf(f, "%s overall alignment rate\n", pct(nuni + nrep, nread).c_str());
	}
};

// Alignment summary of a paired run (AlnSink::printAlSumm, aln_sink.cpp:377-528)
struct PairSummary {
	uint64_t npair = 0, conc0 = 0, conc_uni1 = 0, conc_uni2 = 0, conc_rep = 0, ndiscord = 0;
	uint64_t unp00 = 0, unp0_uni1 = 0, unp0_uni2 = 0, unp0_rep = 0;
	void merge(const PairSummary& o) {
		npair += o.npair; conc0 += o.conc0; conc_uni1 += o.conc_uni1; conc_uni2 += o.conc_uni2; conc_rep += o.conc_rep; ndiscord += o.ndiscord;
		unp00 += o.unp00; unp0_uni1 += o.unp0_uni1; unp0_uni2 += o.unp0_uni2; unp0_rep += o.unp0_rep;
	}
	void add(const ReadResult& r1, const ReadResult& r2) {
		npair++;
		if (r1.pair_type == 1) {
			if (r1.pair_flags & 1) conc_rep++; else if (!(r1.pair_flags & 2)) conc_uni1++; else conc_uni2++;
			return;
		}
		conc0++;
		if (r1.pair_type == 2) { ndiscord++; return; }
		const ReadResult* rr[2] = {&r1, &r2};
		for (int m = 0; m < 2; m++) {
			if (rr[m]->aligned) { if (rr[m]->maxed) unp0_rep++; else if (rr[m]->nalns == 1) unp0_uni1++; else unp0_uni2++; }
			else if (rr[m]->maxed) unp0_rep++;
			else unp00++;
		}
	}
	void print(FILE* f, bool discord, bool mixed) const {
		auto pct = [](uint64_t a, uint64_t b) { char buf[32]; snprintf(buf, sizeof buf, "%.2f%%", b ? 100.0 * (double)a / (double)b : 0.0); return std::string(buf); };
		auto L = [](uint64_t v) { return (unsigned long long)v; };
		if (npair > 0) fprintf(f, "%llu reads; of these:\n", L(npair)); else fprintf(f, "0 reads\n");
		if (npair > 0) {
			fprintf(f, "  %llu (%s) were paired; of these:\n", L(npair), pct(npair, npair).c_str());
			fprintf(f, "    %llu (%s) aligned concordantly 0 times\n", L(conc0), pct(conc0, npair).c_str());
			fprintf(f, "    %llu (%s) aligned concordantly exactly 1 time\n", L(conc_uni1), pct(conc_uni1, npair).c_str());
			fprintf(f, "    %llu (%s) aligned concordantly >1 times\n", L(conc_uni2 + conc_rep), pct(conc_uni2 + conc_rep, npair).c_str());
			if (discord) {
				fprintf(f, "    ----\n");
				fprintf(f, "    %llu pairs aligned concordantly 0 times; of these:\n", L(conc0));
				fprintf(f, "      %llu (%s) aligned discordantly 1 time\n", L(ndiscord), pct(ndiscord, conc0).c_str());
			}
			const uint64_t ncd0 = conc0 - ndiscord;
			if (mixed) {
				fprintf(f, "    ----\n");
				fprintf(f, "    %llu pairs aligned 0 times concordantly or discordantly; of these:\n", L(ncd0));
				fprintf(f, "      %llu mates make up the pairs; of these:\n", L(ncd0 * 2));
				fprintf(f, "        %llu (%s) aligned 0 times\n", L(unp00), pct(unp00, ncd0 * 2).c_str());
				fprintf(f, "        %llu (%s) aligned exactly 1 time\n", L(unp0_uni1), pct(unp0_uni1, ncd0 * 2).c_str());
				fprintf(f, "        %llu (%s) aligned >1 times\n", L(unp0_uni2 + unp0_rep), pct(unp0_uni2 + unp0_rep, ncd0 * 2).c_str());
			}
		}
		const uint64_t tot_al = (conc_uni1 + conc_uni2 + conc_rep) * 2 + ndiscord * 2 + unp0_uni1 + unp0_uni2 + unp0_rep;
		fprintf(f, "%s overall alignment rate\n", pct(tot_al, npair * 2).c_str());
	}
};

// Alignment summary of a run that had pairs AND unpaired reads (AlnSink::printAlSumm, aln_sink.cpp:349-560): one "reads" total in which a
// pair counts once, both sections with their share of it, one overall rate over mates + unpaired reads.
inline void print_mixed_summary(FILE* f, const PairSummary& p, const AlnSummary& u, bool discord, bool mixed) {
	auto pct = [](uint64_t a, uint64_t b) { char buf[32]; snprintf(buf, sizeof buf, "%.2f%%", b ? 100.0 * (double)a / (double)b : 0.0); return std::string(buf); };
	auto L = [](uint64_t v) { return (unsigned long long)v; };
	const uint64_t tot = p.npair + u.nread;
	if (tot > 0) fprintf(f, "%llu reads; of these:\n", L(tot)); else fprintf(f, "0 reads\n");
	if (p.npair > 0) {
		fprintf(f, "  %llu (%s) were paired; of these:\n", L(p.npair), pct(p.npair, tot).c_str());
		fprintf(f, "    %llu (%s) aligned concordantly 0 times\n", L(p.conc0), pct(p.conc0, p.npair).c_str());
		fprintf(f, "    %llu (%s) aligned concordantly exactly 1 time\n", L(p.conc_uni1), pct(p.conc_uni1, p.npair).c_str());
		fprintf(f, "    %llu (%s) aligned concordantly >1 times\n", L(p.conc_uni2 + p.conc_rep), pct(p.conc_uni2 + p.conc_rep, p.npair).c_str());
		if (discord) {
			fprintf(f, "    ----\n");
			fprintf(f, "    %llu pairs aligned concordantly 0 times; of these:\n", L(p.conc0));
			fprintf(f, "      %llu (%s) aligned discordantly 1 time\n", L(p.ndiscord), pct(p.ndiscord, p.conc0).c_str());
		}
		const uint64_t ncd0 = p.conc0 - p.ndiscord;
		if (mixed) {
			fprintf(f, "    ----\n");
			fprintf(f, "    %llu pairs aligned 0 times concordantly or discordantly; of these:\n", L(ncd0));
			fprintf(f, "      %llu mates make up the pairs; of these:\n", L(ncd0 * 2));
			fprintf(f, "        %llu (%s) aligned 0 times\n", L(p.unp00), pct(p.unp00, ncd0 * 2).c_str());
			fprintf(f, "        %llu (%s) aligned exactly 1 time\n", L(p.unp0_uni1), pct(p.unp0_uni1, ncd0 * 2).c_str());
			fprintf(f, "        %llu (%s) aligned >1 times\n", L(p.unp0_uni2 + p.unp0_rep), pct(p.unp0_uni2 + p.unp0_rep, ncd0 * 2).c_str());
		}
	}
	if (u.nread > 0) {
		fprintf(f, "  %llu (%s) were unpaired; of these:\n", L(u.nread), pct(u.nread, tot).c_str());
		fprintf(f, "    %llu (%s) aligned 0 times\n", L(u.n0), pct(u.n0, u.nread).c_str());
		fprintf(f, "    %llu (%s) aligned exactly 1 time\n", L(u.nuni), pct(u.nuni, u.nread).c_str());
		fprintf(f, "    %llu (%s) aligned >1 times\n", L(u.nrep), pct(u.nrep, u.nread).c_str());
	}
	const uint64_t cand = u.nread + p.npair * 2;
	const uint64_t al = (p.conc_uni1 + p.conc_uni2 + p.conc_rep) * 2 + p.ndiscord * 2 + p.unp0_uni1 + p.unp0_uni2 + p.unp0_rep + u.nuni + u.nrep;
	fprintf(f, "%s overall alignment rate\n", pct(al, cand).c_str());
}

} // namespace bt2g
#endif
