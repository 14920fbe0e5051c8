// bt2g_cli.hpp -- command-line options of bowtie2-align-{s,l} that this build implements (bt2_search.cpp:1040-1620).
// Shared by the drop-in binary and by the host-compiled test harness so both parse argv the same way.  Anything
// outside the implemented path is refused with a message (never silently approximated).
#ifndef BT2G_CLI_HPP_
#define BT2G_CLI_HPP_

#include <cstdio>
#include <sys/types.h>
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>

#include "bt2g_host.hpp"

namespace bt2g {

struct CliExtra {
	std::vector<int> devices;      // --gpu a[,b,...]
	bool metrics = false;          // --met: per-read work counters on stderr (test aid)
	bool arg_desc = false;         // --arg-desc
	size_t batch_reads = 1u << 18;
	size_t batch_max = 0;              // --batch-max N: vary the batch size instead (ramp up from 64 K reads to N, taper off towards the end of a plain file); see bt2g_search.cpp
	bool version = false, help = false;   // --version / -h
	bool allow_paired = false;     // set by the caller before parsing: this front end can run pairs
	// --shard r/N: this process is rank r of N (one process per GPU).  The input is cut into blocks of --batch
	// reads (pairs stay together) dealt round-robin; the process aligns blocks r, r+N, ... only, writes their SAM in block order
	// and lists "block_id bytes" per block plus its summary counters in --shard-index FILE, from which rank 0 reassembles
	// the ordered SAM (bowtie2_amd/mgpu.py gathers the pieces over RCCL).
	int shard_rank = 0, shard_world = 1;
	std::string shard_index;
	// --shard-bytes a:b[,a2:b2]: this process aligns the records in bytes [a, b) of the -U file (pairs: of the -1 file, and [a2, b2)
	// of the -2 file) and nothing else -- it never reads another rank's bytes.  The ranges are record-aligned and computed by the
	// launcher (bowtie2_amd/mgpu.py: newline counts of every rank's slice, all-gathered).  --shard-first-read K: number of the first
	// read / pair of the range within the whole input (reads without a name are named by their number).  The SAM pieces of ranks
	// 0..N-1 concatenate to the one-process output; the summary counters go to --shard-index as for --shard.
	bool shard_bytes = false;
	uint64_t range_a[2] = {0, 0}, range_b[2] = {0, 0};
	int n_ranges = 0;
	uint64_t first_read = 0;
	std::string pg_cmdline;        // --pg-cmdline TEXT: the command line to show in @PG (the launcher passes the user's own)
};

inline bool split_ints(const std::string& s, char sep, std::vector<int>& out) {
	out.clear();
	size_t p = 0;
	while (p <= s.size()) {
		size_t q = s.find(sep, p);
		if (q == std::string::npos) q = s.size();
		if (q == p) return false;
		char* end = nullptr;
		const std::string t = s.substr(p, q - p);
		const long v = strtol(t.c_str(), &end, 10);
		if (*end) return false;
		out.push_back((int)v);
		p = q + 1;
	}
	return !out.empty();
}

// The option interface of bowtie2-align (name:takes-argument), in the order `--arg-desc` lists it
// (printArgDesc, bt2_search.cpp:710-747).  The Perl wrapper asks for this table (`--wrapper basic-0 --arg-desc`,
// bowtie2:104) to tell the aligner's options from its own, so a drop-in has to answer with the same names --
// including the options this build goes on to reject.
static const char kArgDesc[] =
	"verbose:0 startverbose:0 quiet:0 sanity:0 pause:0 orig:1 all:0 solexa-quals:0 integer-quals:0 "
	"int-quals:0 metrics:1 metrics-file:1 metrics-stderr:0 metrics-per-read:0 met-read:0 met:1 met-file:1 "
	"met-stderr:0 time:0 trim3:1 trim5:1 seed:1 qupto:1 upto:1 version:0 reads-per-batch:1 filepar:0 "
	"help:0 threads:1 khits:1 lowseeds:1 minins:1 maxins:1 quals:1 Q1:1 Q2:1 refidx:0 partition:1 ff:0 "
	"fr:0 rf:0 cachelim:1 cachesz:1 nofw:0 norc:0 skip:1 12:1 tab5:1 tab6:1 interleaved:1 phred33-quals:0 "
	"phred64-quals:0 phred33:0 phred64:0 solexa1.3-quals:0 mm:0 shmem:0 mmsweep:0 hadoopout:0 fullref:0 "
	"usage:0 sam-no-qname-trunc:0 sam-omit-sec-seq:0 omit-sec-seq:0 sam-no-head:0 sam-nohead:0 sam-noHD:0 "
	"sam-no-hd:0 sam-nosq:0 sam-no-sq:0 sam-noSQ:0 no-head:0 no-hd:0 no-sq:0 no-HD:0 no-SQ:0 no-unal:0 "
	"sam-RG:1 sam-rg:1 sam-rg-id:1 RG:1 rg:1 rg-id:1 snpphred:1 snpfrac:1 gbar:1 qseq:0 policy:1 preset:1 "
	"seed-summ:0 seed-summary:0 overhang:0 no-cache:0 cache:0 454:0 ion-torrent:0 no-mixed:0 "
	"no-discordant:0 local:0 end-to-end:0 ungapped:0 no-ungapped:0 sse8:0 no-sse8:0 scan-narrowed:0 "
	"qc-filter:0 bwa-sw-like:0 multiseed:1 ma:1 mp:1 np:1 rdg:1 rfg:1 score-min:1 min-score:1 n-ceil:1 "
	"dpad:1 mapq-print-inputs:0 very-fast:0 fast:0 sensitive:0 very-sensitive:0 very-fast-local:0 "
	"fast-local:0 sensitive-local:0 very-sensitive-local:0 seedlen:1 seedmms:1 seedival:1 ignore-quals:0 "
	"index:1 arg-desc:0 wrapper:1 unpaired:1 output:1 mapq-v:1 dovetail:0 no-dovetail:0 contain:0 "
	"no-contain:0 overlap:0 no-overlap:0 tighten:1 exact-upfront:0 1mm-upfront:0 no-exact-upfront:0 "
	"no-1mm-upfront:0 1mm-minlen:1 deterministic-seeds:0 no-deterministic-seeds:0 seed-off:1 seed-boost:1 "
	"read-times:0 show-rand-seed:0 dp-fail-streak:1 ee-fail-streak:1 ug-fail-streak:1 fail-streak:1 "
	"dp-fails:1 ug-fails:1 extends:1 no-extend:0 mapq-extra:0 seed-rounds:1 reorder:0 passthrough:0 "
	"sample:1 cp-min:1 cp-ival:1 tri:0 nondeterministic:0 non-deterministic:0 local-seed-cache-sz:1 "
	"seed-cache-sz:1 no-unal:0 test-25:0 desc-kb:1 desc-landing:1 desc-exp:1 desc-prioritize:0 "
	"desc-fmops:1 log-dp:1 log-dp-opp:1 soft-clipped-unmapped-tlen:0 xeq:0 thread-ceiling:1 "
	"thread-piddir:1 trim-to:1 preserve-tags:0 align-paired-reads:0 sam-append-comment:0 sam-opt-config:1 "
	"b:0 f:0 F:1 q:0 b:0 z:0 h:0 c:0 u:1 r:0 v:1 s:1 a:0 d:0 P:1 t:0 3:1 5:1 w:1 p:1 k:1 l:1 M:1 1:1 2:1 "
	"I:1 X:1 C:0 Q:1 N:1 i:1 L:1 U:1 x:1 S:1 g:1 O:1 D:1 R:1 ";

inline void print_arg_desc() {
	std::string tok;
	for (const char* p = kArgDesc; ; p++) {
		if (*p == ' ' || *p == 0) {
			const size_t c = tok.rfind(':');
			if (c != std::string::npos) printf("%s\t%s\n", tok.substr(0, c).c_str(), tok.substr(c + 1).c_str());
			tok.clear();
			if (*p == 0) break;
		} else tok.push_back(*p);
	}
}

// --version in the reference's layout (bt2_search.cpp:5296-5320: "<argv0> version <v>" first, which is what wrappers and report
// tools parse); the version is the reference release whose behaviour this build reproduces
inline void print_version(const char* argv0) {
	printf("%s version 2.5.5\n64-bit\nBuilt for AMD MI355X (gfx950): bowtie2_amd, the multiseed hot path on the device; no CPU alignment path\n", argv0);
	printf("Sizeof {int, long, long long, void*, size_t, off_t}: {%zu, %zu, %zu, %zu, %zu, %zu}\n", sizeof(int), sizeof(long), sizeof(long long), sizeof(void*), sizeof(size_t), sizeof(off_t));
}
inline void print_usage(const char* argv0) {
	printf("Usage: %s [options] -x <bt2-idx> {-1 <m1> -2 <m2> | -U <r> | --interleaved <i> | --tab5/--tab6 <f>} [-S <sam>]   (FASTQ: -U may accompany -1/-2)\n", argv0);
	printf("  inputs: -q -f -r -c --qseq (plain or gzipped, comma-separated lists)  -b <unaligned BAM> [--align-paired-reads] [--preserve-tags]\n"
	       "          -s/-u -5/-3 --trim-to --phred33/--phred64/--solexa-quals\n"
	       "  presets: --very-fast --fast --sensitive --very-sensitive (and -local)   --end-to-end | --local\n"
	       "  alignment: -N 0|1 -L -i --n-ceil --dpad --gbar --ignore-quals --nofw --norc --no-1mm-upfront --no-exact-upfront -d --overhang\n"
	       "  scoring: --ma --mp --np --rdg --rfg --score-min --policy --bwa-sw-like      effort: -D -R --extends --dp-fails --ug-fails --seed-boost --tighten --no-extend --[no-]ungapped\n"
	       "  reporting: -k <=1000 | -a | -M\n"
	       "  pairs: -I -X --fr/--rf/--ff --no-mixed --no-discordant --dovetail --no-contain --no-overlap\n"
	       "  SAM: --no-unal --no-hd --no-sq --rg-id --rg --omit-sec-seq --sam-no-qname-trunc --sam-append-comment --soft-clipped-unmapped-tlen --xeq --passthrough\n"
	       "  other: -p --reorder -t --quiet --seed --qc-filter --gpu a,b --batch n\n"
	       "  Options of bowtie2 outside this list are refused, never approximated.\n");
}

// One ';'-separated TAG=VALUE list in the reference's internal policy syntax (--policy; SeedAlignmentPolicy::parseString,
// aligner_seed_policy.cpp:245-640).  Returns "" or the error text.
inline std::string apply_policy_string(Options& opt, const std::string& pol) {
	size_t p = 0;
	while (p <= pol.size()) {
		size_t q = pol.find(';', p);
		if (q == std::string::npos) q = pol.size();
		const std::string tok = pol.substr(p, q - p);
		p = q + 1;
		if (tok.empty()) continue;
		const size_t eq = tok.find('=');
		if (eq == std::string::npos || tok.find('=', eq + 1) != std::string::npos) return "Error parsing alignment policy setting; must be bisected by = sign";
		const std::string tag = tok.substr(0, eq), val = tok.substr(eq + 1);
		std::vector<std::string> c;
		for (size_t a = 0; a <= val.size();) { size_t b = val.find(',', a); if (b == std::string::npos) b = val.size(); c.push_back(val.substr(a, b - a)); a = b + 1; }
		if (val.empty() || c.empty()) return "Error parsing alignment policy setting; RHS must have at least 1 token";
		for (const std::string& t : c) if (t.empty()) return "Error parsing alignment policy setting; empty token on RHS";
		if (tag == "MA") { opt.ma = atoi(c[0].c_str()); opt.set_ma = true; }
		else if (tag == "MMP") {
			if (c.size() > 3) return "MMP: RHS must have at most 3 tokens";
			if (c[0][0] == 'C') { opt.mp_max = opt.mp_min = atoi(c[0].c_str() + 1); opt.mm_const = true; opt.mm_rounded = false; }
			else if (c[0][0] == 'R') { opt.mm_rounded = true; opt.mm_const = false; }
			else if (c[0][0] == 'Q') {
				opt.mp_max = c.size() >= 2 ? atoi(c[1].c_str()) : 6;
				opt.mp_min = c.size() >= 3 ? atoi(c[2].c_str()) : 2;
				if (opt.mp_min > opt.mp_max) return "Maximum mismatch penalty is less than minimum penalty";
				opt.mm_const = false; opt.mm_rounded = false;
			} else return "MMP=" + c[0] + ": RHS must start with C, Q or R";
		}
		else if (tag == "NP") {
			if (c.size() != 1) return "NP: RHS must have 1 token";
			// NP=Q keeps the current constant: the N table is built with min = max = penN, so the quality model degenerates (scoring.h:113-119)
			if (c[0][0] == 'Q') {}
			else if (c[0][0] != 'C') return "NP=" + c[0] + " is not supported by this build (constant or Q only)";
			else opt.np = atoi(c[0].c_str() + 1);
		}
		else if (tag == "RDG") { opt.rdg_const = atoi(c[0].c_str()); opt.rdg_linear = c.size() >= 2 ? atoi(c[1].c_str()) : 3; }
		else if (tag == "RFG") { opt.rfg_const = atoi(c[0].c_str()); opt.rfg_linear = c.size() >= 2 ? atoi(c[1].c_str()) : 3; }
		else if (tag == "MIN") { opt.set_score_min = true; if (!opt.score_min.parse(val)) return "bad MIN function"; }
		else if (tag == "NCEIL") { if (!opt.n_ceil.parse(val)) return "bad NCEIL function"; }
		else if (tag == "SEED") {
			if (c.size() > 1) return "SEED: RHS must have 1 token";
			const int mms = atoi(c[0].c_str());
			if (mms < 0 || mms > 1) return "Error: -N was set to " + c[0] + ", but cannot be set higher than 1 or less than 0";
			opt.seed_mms = mms;
		}
		else if (tag == "SEEDLEN") { opt.seed_len = atoi(c[0].c_str()); opt.set_L = true; }
		else if (tag == "DPS") { opt.max_dp_streak = atoi(c[0].c_str()); opt.set_D = true; }
		else if (tag == "ROUNDS") { opt.n_seed_rounds = atoi(c[0].c_str()); opt.set_R = true; }
		else if (tag == "IVAL") { opt.set_i = true; if (!opt.ms_ival.parse(val)) return "bad IVAL function"; }
		else return "Unexpected alignment policy setting '" + tag + "'";
	}
	return "";
}

// Returns "" on success, else the error text (exit code 1).
inline std::string parse_cli(int argc, char** argv, Options& opt, CliExtra& ex) {
	bool saw_bam = false;
	for (int i = 1; i < argc; i++) {
		std::string a = argv[i];
		std::string inline_val;
		bool has_inline = false;
		if (a.size() > 2 && a[0] == '-' && a[1] == '-') {           // --opt=value
			const size_t eq = a.find('=');
			if (eq != std::string::npos) { inline_val = a.substr(eq + 1); a = a.substr(0, eq); has_inline = true; }
		}
		// long spellings of the short options (the getopt table, bt2_search.cpp:505-705)
		{
			static const char* const alias[][2] = {{"--khits", "-k"}, {"--seedlen", "-L"}, {"--seedmms", "-N"}, {"--seedival", "-i"}, {"--index", "-x"}, {"--unpaired", "-U"},
				{"--usage", "-h"}, {"--seed-rounds", "-R"}, {"--fail-streak", "-D"}, {"--12", "--tab5"}, {"--minins", "-I"}, {"--maxins", "-X"}};
			for (const auto& al : alias) if (a == al[0]) { a = al[1]; break; }
		}
		std::string err;
		auto need = [&]() -> std::string {
			if (has_inline) return inline_val;
			if (i + 1 >= argc) { err = a + " needs an argument"; return ""; }
			return argv[++i];
		};
		std::vector<int> iv;
		if (a == "--wrapper") { need(); }
		else if (a == "--arg-desc") ex.arg_desc = true;
		else if (a == "--version") { ex.version = true; return ""; }
		else if (a == "-h" || a == "--help") { ex.help = true; return ""; }
		else if (a == "-x") opt.index_base = need();
		else if (a == "-U") opt.reads_file = need();
		else if (a == "-S" || a == "--output") opt.out_file = need();
		else if (a == "-q") opt.format = 0;
		else if (a == "-f") opt.format = 1;
		else if (a == "-r") opt.format = 2;
		else if (a == "-c") opt.format = 3;
		else if (a == "--qseq") opt.format = 5;
		else if (a == "-F") {
			// -F k:<int>,i:<int> (bt2_search.cpp:1104-1112)
			const std::string v = need();
			int k = 0, iv2 = 0;
			if (sscanf(v.c_str(), "%d,%d", &k, &iv2) != 2) err = "-F expects <length>,<interval> (the aligner binary parses a plain pair, bt2_search.cpp:1109)";
			else if (k < 1 || k > 1024 || iv2 < 1) err = "-F: k must be in [1, 1024] and i positive";
			else { opt.format = 6; opt.fc_len = k; opt.fc_freq = iv2; }
		}
		else if (a == "--qc-filter") opt.qc_filter = true;
		else if (a == "--sam-no-qname-trunc") opt.sam_no_qname_trunc = true;
		else if (a == "--tab5" || a == "--tab6") { opt.format = 4; opt.reads_file = need(); }
		else if (a == "-p" || a == "--threads") opt.threads = atoi(need().c_str());
		else if (a == "--reorder") opt.reorder = true;
		else if (a == "-t" || a == "--time") opt.timing = true;
		else if (a == "--quiet") opt.quiet = true;
		else if (a == "-k") { opt.khits = atoi(need().c_str()); opt.saw_k = true; if (opt.khits < 1) err = "-k arg must be at least 1"; }
		else if (a == "-a" || a == "--all") { opt.all_hits = true; opt.saw_k = false; }
		else if (a == "-M") { opt.mhits = atoi(need().c_str()); opt.saw_k = false; opt.khits = 1; fprintf(stderr, "Warning: -M is deprecated.  Use -D and -R to adjust effort instead.\n"); }
		else if (a == "-s" || a == "--skip") opt.skip = strtoull(need().c_str(), nullptr, 10);
		else if (a == "-u" || a == "--upto" || a == "--qupto") { opt.upto = strtoull(need().c_str(), nullptr, 10); if (opt.upto == 0) opt.upto = UINT64_MAX; }
		else if (a == "-5" || a == "--trim5") opt.trim5 = atoi(need().c_str());
		else if (a == "-3" || a == "--trim3") opt.trim3 = atoi(need().c_str());
		else if (a == "--phred33" || a == "--phred33-quals") { opt.phred64 = false; opt.solexa_quals = false; }      // bt2_search.cpp:1176
		else if (a == "--phred64" || a == "--phred64-quals" || a == "--solexa1.3-quals") opt.phred64 = true;
		else if (a == "--solexa-quals") opt.solexa_quals = true;
		else if (a == "--seed") opt.seed = (uint32_t)strtoul(need().c_str(), nullptr, 10);
		else if (a == "--nofw") opt.nofw = true;
		else if (a == "--norc") opt.norc = true;
		else if (a == "--end-to-end") opt.local = false;
		else if (a == "--ignore-quals") opt.ignore_quals = true;
		else if (a == "--no-1mm-upfront") opt.no_1mm_upfront = true;
		else if (a == "--1mm-upfront") opt.no_1mm_upfront = false;
		else if (a == "--no-exact-upfront") opt.no_exact_upfront = true;
		else if (a == "--exact-upfront") opt.no_exact_upfront = false;
		else if (a == "-d" || a == "--deterministic-seeds") opt.det_seeds = true;
		else if (a == "--no-deterministic-seeds") opt.det_seeds = false;
		else if (a == "--seed-cache-sz") { opt.seed_cache_mb = atoi(need().c_str()); if (opt.seed_cache_mb < 1) err = "--seed-cache-sz arg must be at least 1"; }
		else if (a == "--local-seed-cache-sz") { (void)need(); }      // across-read cache: not used by the reference either (msNoCache)
		else if (a == "--no-unal") opt.no_unal = true;
		else if (a == "--xeq") opt.xeq = true;
		else if (a == "--omit-sec-seq" || a == "--sam-omit-sec-seq") opt.omit_sec_seq = true;
		else if (a == "--no-hd" || a == "--no-head" || a == "--sam-no-hd" || a == "--sam-nohead" || a == "--sam-no-head" || a == "--sam-noHD" || a == "--no-HD") opt.sam_no_hd = true;
		else if (a == "--no-sq" || a == "--sam-no-sq" || a == "--sam-nosq" || a == "--sam-noSQ" || a == "--no-SQ") opt.sam_no_sq = true;
		else if (a == "--rg-id" || a == "--sam-rg-id") { const std::string v = need(); opt.rg_id = "\tID:" + v; opt.rg_optflag = "RG:Z:" + v; }
		else if (a == "--rg" || a == "--sam-rg" || a == "--sam-RG" || a == "--RG") {
			const std::string v = need();
			if (v.substr(0, 3) == "ID:") { opt.rg_id = "\t" + v; opt.rg_optflag = "RG:Z:" + v.substr(3); } else { opt.rgs += "\t" + v; }
		}
		else if (a == "--gpu") { if (!split_ints(need(), ',', ex.devices)) err = "--gpu needs a comma-separated list of device indexes"; }
		else if (a == "--met") ex.metrics = true;
		else if (a == "--batch") ex.batch_reads = strtoull(need().c_str(), nullptr, 10);
		else if (a == "--batch-max") ex.batch_max = strtoull(need().c_str(), nullptr, 10);
		else if (a == "--shard") {
			const std::string v = need();
			if (sscanf(v.c_str(), "%d/%d", &ex.shard_rank, &ex.shard_world) != 2 || ex.shard_world < 1 || ex.shard_rank < 0 || ex.shard_rank >= ex.shard_world) err = "--shard needs r/N with 0 <= r < N";
		}
		else if (a == "--shard-index") ex.shard_index = need();
		else if (a == "--shard-bytes") {
			const std::string v = need();
			unsigned long long x[4] = {0, 0, 0, 0};
			const int k = sscanf(v.c_str(), "%llu:%llu,%llu:%llu", &x[0], &x[1], &x[2], &x[3]);
			if (k != 2 && k != 4) err = "--shard-bytes needs a:b or a:b,a2:b2";
			else { ex.shard_bytes = true; ex.n_ranges = k / 2; for (int j = 0; j < ex.n_ranges; j++) { ex.range_a[j] = x[2 * j]; ex.range_b[j] = x[2 * j + 1]; } }
		}
		else if (a == "--shard-first-read") ex.first_read = strtoull(need().c_str(), nullptr, 10);
		else if (a == "--pg-cmdline") ex.pg_cmdline = need();
		else if (a == "-D") { opt.max_dp_streak = atoi(need().c_str()); opt.set_D = true; }
		else if (a == "-R") { opt.n_seed_rounds = atoi(need().c_str()); opt.set_R = true; }
		else if (a == "-L") { opt.seed_len = atoi(need().c_str()); opt.set_L = true; if (opt.seed_len < 1 || opt.seed_len > 32) err = "-L argument must be in [1, 32]"; }
		else if (a == "--local") opt.local = true;
		else if (a == "--overhang") opt.report_overhangs = true;
		else if (a == "--passthrough") opt.passthrough = true;
		else if (a == "--policy") err = apply_policy_string(opt, need());
		else if (a == "--bwa-sw-like") {
			// bt2_search.cpp:1114-1126: local mode, BWA-SW's scoring, and its length-dependent score threshold
			opt.local = true; opt.bwa_sw_like = true;
			err = apply_policy_string(opt, "MA=1;MMP=C3;RDG=5,2;RFG=5,2");
		}
		// paired-end input and policy (bt2_search.cpp:1185-1215); without -1/-2 the policy options have no effect
		else if (a == "--interleaved") opt.interleaved_file = need();
		else if (a == "-1") opt.mate1_file = need();
		else if (a == "-2") opt.mate2_file = need();
		else if (a == "-I" || a == "--minins") opt.min_insert = atoi(need().c_str());
		else if (a == "-X" || a == "--maxins") opt.max_insert = atoi(need().c_str());
		else if (a == "--fr") { opt.mate1fw = true; opt.mate2fw = false; }
		else if (a == "--rf") { opt.mate1fw = false; opt.mate2fw = true; }
		else if (a == "--ff") { opt.mate1fw = true; opt.mate2fw = true; }
		else if (a == "--no-mixed") opt.no_mixed = true;
		else if (a == "--no-discordant") opt.no_discordant = true;
		else if (a == "--dovetail") opt.dovetail = true;
		else if (a == "--no-dovetail") opt.dovetail = false;
		else if (a == "--no-contain") opt.no_contain = true;
		else if (a == "--no-overlap") opt.no_overlap = true;
		else if (a == "--contain") opt.no_contain = false;
		else if (a == "--overlap") opt.no_overlap = false;
		// effort knobs (bt2_search.cpp:1274-1310, 1461-1477)
		else if (a == "--extends") { opt.max_iters = atoi(need().c_str()); if (opt.max_iters < 0) err = "--extends must not be negative"; }
		else if (a == "--dp-fails") { opt.max_dp = atoi(need().c_str()); if (opt.max_dp < 0) err = "--dp-fails must not be negative"; }
		else if (a == "--ug-fails") { opt.max_ug = atoi(need().c_str()); if (opt.max_ug < 0) err = "--ug-fails must not be negative"; }
		else if (a == "--seed-boost") { opt.seed_boost_thresh = atoi(need().c_str()); if (opt.seed_boost_thresh < 0) err = "--seed-boost must not be negative"; }
		else if (a == "--tighten") opt.tighten = atoi(need().c_str());
		else if (a == "--no-extend") opt.do_extend = false;
		else if (a == "--ungapped") opt.do_ungapped = true;
		else if (a == "--no-ungapped") opt.do_ungapped = false;
		// accepted and without effect on the output, here as in the reference: batching / thread-pool housekeeping of the CPU program, and
		// --1mm-minlen, which bt2_search.cpp parses (:1477) and never reads
		else if (a == "--reads-per-batch" || a == "--thread-ceiling" || a == "--thread-piddir" || a == "--1mm-minlen") { (void)need(); }
		else if (a == "-N") { const std::string v = need(); opt.seed_mms = atoi(v.c_str()); if (opt.seed_mms < 0 || opt.seed_mms > 1) err = "Error: -N was set to " + v + ", but cannot be set higher than 1 or less than 0"; }
		else if (a == "-i") { opt.set_i = true; if (!opt.ms_ival.parse(need())) err = "bad -i function"; }
		else if (a == "--score-min" || a == "--min-score") { opt.set_score_min = true; if (!opt.score_min.parse(need())) err = "bad --score-min function"; }
		else if (a == "--n-ceil") {
			// 3 tokens: a function; 1-2 tokens: linear, "L,<const>[,<coeff>]" (bt2_search.cpp:1565-1588)
			std::string v = need();
			size_t ncomma = 0; for (char ch : v) if (ch == ',') ncomma++;
			if (ncomma < 2) v = "L," + v;
			if (ncomma > 2 || !opt.n_ceil.parse(v)) err = "bad --n-ceil function";
		}
		else if (a == "--multiseed") {
			// N,L,F,C,L: seed mismatches, seed length, interval function (bt2_search.cpp:1545-1564)
			const std::string v = need();
			std::vector<std::string> t; size_t p0 = 0;
			while (true) { const size_t q = v.find(',', p0); t.push_back(v.substr(p0, q == std::string::npos ? q : q - p0)); if (q == std::string::npos) break; p0 = q + 1; }
			if (t.empty() || t.size() > 5 || t[0].empty()) err = "expected 5 or fewer comma-separated arguments to --multiseed";
			else {
				opt.seed_mms = atoi(t[0].c_str());
				if (opt.seed_mms < 0 || opt.seed_mms > 1) err = "Error: -N was set to " + t[0] + ", but cannot be set higher than 1 or less than 0";
				if (t.size() > 1) { opt.seed_len = atoi(t[1].c_str()); opt.set_L = true; if (opt.seed_len < 1 || opt.seed_len > 32) err = "-L argument must be in [1, 32]"; }
				if (t.size() > 2) {
					std::string f = t[2];
					for (size_t k = 3; k < t.size(); k++) f += "," + t[k];
					opt.set_i = true;
					if (!opt.ms_ival.parse(f)) err = "bad --multiseed interval function";
				}
			}
		}
		else if (a == "--trim-to") {
			const std::string v = need();
			const size_t colon = v.find(':');
			if (colon == std::string::npos) { opt.trim_to_side = 3; opt.trim_to_len = atoi(v.c_str()); }
			else { opt.trim_to_side = atoi(v.substr(0, colon).c_str()); opt.trim_to_len = atoi(v.substr(colon + 1).c_str()); }
			if ((opt.trim_to_side != 3 && opt.trim_to_side != 5) || opt.trim_to_len < 0) err = "--trim-to: trim position must be either 3 or 5 and the length non-negative";
		}
		else if (a == "--dpad") opt.maxhalf = atoi(need().c_str());
		else if (a == "--gbar") { opt.gbar = atoi(need().c_str()); if (opt.gbar < 1) err = "--gbar must be no less than 1"; }
		else if (a == "--ma") { opt.ma = atoi(need().c_str()); opt.set_ma = true; }
		else if (a == "--mp") {
			if (!split_ints(need(), ',', iv) || iv.size() > 2) err = "expected 1 or 2 comma-separated arguments to --mp";
			else { opt.mp_max = iv[0]; opt.mp_min = iv.size() > 1 ? iv[1] : 2; opt.mm_const = false; opt.mm_rounded = false;    // "MMP=Q,max[,min]" appended to the policy string (bt2_search.cpp:1591-1608): a later --mp undoes an earlier MMP=C
			       if (opt.mp_min > opt.mp_max) err = "Maximum mismatch penalty is less than minimum penalty"; }
		}
		else if (a == "--np") opt.np = atoi(need().c_str());
		else if (a == "--rdg") { if (!split_ints(need(), ',', iv) || iv.size() > 2) err = "bad --rdg"; else { opt.rdg_const = iv[0]; opt.rdg_linear = iv.size() > 1 ? iv[1] : 3; } }   // a missing extension penalty falls back to the default (aligner_seed_policy.cpp:490-500)
		else if (a == "--rfg") { if (!split_ints(need(), ',', iv) || iv.size() > 2) err = "bad --rfg"; else { opt.rfg_const = iv[0]; opt.rfg_linear = iv.size() > 1 ? iv[1] : 3; } }
		else if (a.size() > 2 && a.substr(0, 2) == "--" && !has_inline && opt.apply_preset(a.substr(2))) {}
		else if (a == "--int-quals" || a == "--integer-quals")
			// the reference's own parser folds the separating blank into the next number and aborts with "Saw negative Phred quality"
			// on any record with two or more values (pat.cpp:1191-1204); there is no behaviour to reproduce
			return "option " + a + " is not supported (bowtie2's parser aborts on space-separated integer qualities)";
		else if (a == "-b") { opt.format = 7; saw_bam = true; }
		else if (a == "--preserve-tags") { opt.preserve_tags = true; }
		else if (a == "--align-paired-reads") { opt.align_paired_reads = true; }
		else if (a == "--sam-append-comment") opt.sam_append_comment = true;
		else if (a == "--soft-clipped-unmapped-tlen") opt.sc_unmapped = true;
		else return "unsupported option " + a;
		if (!err.empty()) return err;
	}
	if (opt.khits > BT2G_MAX_KHITS) return "-k above 1000 is not supported by this build";
	// bt2_search.cpp:1699-1718, 1804-1807
	if (!opt.local && opt.sc_unmapped) return "ERROR: --soft-clipped-unmapped-tlen can only be set for local alignments.";
	if (!saw_bam && opt.preserve_tags) return "--preserve_tags can only be used when aligning BAM reads.";
	if (!saw_bam && opt.align_paired_reads) return "--align-paired-reads can only be used when aligning BAM reads.";
	if (opt.sam_append_comment && opt.format != 0 && opt.format != 1) return "Error --sam-append-comment only works with FASTA (-f) and FASTQ (-q) formats. ";
	if (opt.format == 7 && !opt.interleaved_file.empty()) return "-b with --interleaved is not supported by this build (give the BAM file to -1 and -2 with --align-paired-reads)";
	if (opt.det_seeds) {      // bt2_search.cpp:1778-1791
		if (!opt.no_exact_upfront || !opt.no_1mm_upfront) return "Error: -d must be used with --no-exact-upfront and --no-1mm-upfront.";
		if (!opt.all_hits) return "Error: -d can only be used with -a.";
	}
	if (opt.format == 4 && !opt.reads_file.empty()) {
		// --tab5/--tab6: a file of 5- or 6-field records holds pairs (pat.cpp:1545); it is then read like an interleaved source
		FILE* tf = fopen(opt.reads_file.c_str(), "rb");
		if (tf) {
			char line[1 << 16];
			while (fgets(line, sizeof line, tf)) {
				int tabs = 0; bool blank = true;
				for (const char* q = line; *q; q++) { if (*q == '\t') tabs++; if (*q != '\n' && *q != '\r' && *q != ' ') blank = false; }
				if (blank) continue;
				if (tabs >= 4) { opt.interleaved_file = opt.reads_file; opt.reads_file.clear(); }
				break;
			}
			fclose(tf);
		}
	}
	if (opt.mate1_file.empty() != opt.mate2_file.empty()) return "-1 and -2 must be specified together";
	opt.paired = !opt.mate1_file.empty() || !opt.interleaved_file.empty();
	if (!opt.interleaved_file.empty() && !opt.mate1_file.empty()) return "--interleaved and -1/-2 in one run are not supported by this build";
	if (opt.paired && !opt.reads_file.empty()) {
		// -U next to -1/-2: the pair sources are read to their end, then the unpaired ones (PatternComposer, pat.cpp:225-420); one summary
		// FASTQ only: with FASTA input the reference loses the first unpaired record after the pairs, with BAM input its -U source takes paired
		// records (observed on 2.5.5) -- behaviour this build does not reproduce, so those combinations stay refused
		if (opt.format != 0) return "mixing paired and unpaired inputs in one run is supported for FASTQ input only in this build";
		// (-s/-u count within each source, as in the reference: every PatternSource numbers its own reads)
		if (ex.shard_world > 1) return "--shard together with mixed paired and unpaired inputs is not supported by this build";
		opt.mixed_unpaired = true;
	}
	if (ex.shard_bytes) {
		if (opt.format != 0 || !opt.interleaved_file.empty() || opt.mixed_unpaired || opt.skip > 0 || opt.upto != std::numeric_limits<uint64_t>::max() || ex.shard_world > 1)
			return "--shard-bytes is for plain FASTQ files given with -U or -1/-2, without -s/-u, --interleaved, mixed input or --shard";
		if (ex.n_ranges != (opt.paired ? 2 : 1)) return "--shard-bytes needs one byte range per reads file";
	}
	if (opt.paired && !ex.allow_paired) return "paired-end input (-1/-2) is not enabled in this build of the device path yet";
	if (opt.paired && opt.max_insert < opt.min_insert) return "-X must not be smaller than -I";
	if (opt.trim_to_len >= 0 && (opt.trim5 > 0 || opt.trim3 > 0)) return "--trim-to and -3/-5 are mutually exclusive";
	if (opt.set_ma && !opt.local && opt.ma != 0) fprintf(stderr, "Warning: Match bonus always = 0 in --end-to-end mode; ignoring user setting\n");
	if (opt.local && opt.set_ma && opt.ma <= 0) return "--local needs a positive --ma in this build";
	opt.resolve_preset();
	return "";
}

} // namespace bt2g
#endif
