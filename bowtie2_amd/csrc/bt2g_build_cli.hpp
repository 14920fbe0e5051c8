// bt2g_build_cli.hpp -- command line of bowtie2-build-{s,l} (bt2_build.cpp:105-330) for the options that affect the
// index files; memory-tuning options of the reference's blockwise sorter are accepted and ignored (they cannot change
// the output).  Shared by the product binary (device backend) and the CPU twin the tests compile.
#ifndef BT2G_BUILD_CLI_HPP_
#define BT2G_BUILD_CLI_HPP_

#include "bt2g_build_io.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

namespace bt2g { namespace build {

inline double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct CliOpts {
	bool large = false, cmdline_seqs = false, quiet = false, verbose = false, just_ref = false, no_ref = false;
	int off_rate = 4, ftab_chars = 10, line_rate = -1, gpu = 0;
	std::string infile, outfile;
};

// returns 0 to go on, 1 = done (help/version), 2 = usage error
inline int parse_build_args(int argc, const char** argv, bool large_default, CliOpts& o) {
	o.large = large_default;
	std::vector<std::string> pos;
	auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { std::cerr << "Error: option " << argv[i] << " needs an argument" << std::endl; return nullptr; } return argv[++i]; };
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		auto val = [&](const char* lng) -> const char* {      // "--opt v" or "--opt=v"
			const std::string p = std::string(lng) + "=";
			if (a.compare(0, p.size(), p) == 0) return argv[i] + p.size();
			return need(i);
		};
		if (a == "-c") o.cmdline_seqs = true;
		else if (a == "-f") o.cmdline_seqs = false;
		else if (a == "-q" || a == "--quiet") o.quiet = true;
		else if (a == "--verbose") o.verbose = true;
		else if (a == "--large-index") o.large = true;
		else if (a == "-3" || a == "--justref") o.just_ref = true;
		else if (a == "-r" || a == "--noref") o.no_ref = true;
		else if (a == "-o" || a == "--offrate" || a.compare(0, 10, "--offrate=") == 0) { const char* v = a == "-o" ? need(i) : val("--offrate"); if (!v) return 2; o.off_rate = atoi(v); }
		else if (a == "-t" || a == "--ftabchars" || a.compare(0, 12, "--ftabchars=") == 0) { const char* v = a == "-t" ? need(i) : val("--ftabchars"); if (!v) return 2; o.ftab_chars = atoi(v); }
		else if (a == "--gpu" || a.compare(0, 6, "--gpu=") == 0) { const char* v = val("--gpu"); if (!v) return 2; o.gpu = atoi(v); }
		else if (a == "--threads" || a == "--seed" || a == "--bmax" || a == "--bmaxdivn" || a == "--dcv" || a == "-i" || a == "--linesperside" ||
		         a == "--wrapper") { if (!need(i)) return 2; }     // no effect on the files written
		else if (a == "-l" || a == "--linerate") { const char* v = need(i); if (!v) return 2; o.line_rate = atoi(v); }
		else if (a.compare(0, 10, "--threads=") == 0 || a.compare(0, 7, "--seed=") == 0 || a.compare(0, 7, "--bmax=") == 0 || a.compare(0, 11, "--bmaxdivn=") == 0 || a.compare(0, 6, "--dcv=") == 0) {}
		else if (a == "-a" || a == "--noauto" || a == "-p" || a == "--packed" || a == "--nodc" || a == "-s" || a == "--sanity" || a == "--ntoa" || a == "-C" || a == "--color" ||
		         a == "-e" || a == "--entiresa" || a == "--big" || a == "--little") {
			if (a == "--ntoa" || a == "-C" || a == "--color") { std::cerr << "Error: " << a << " is not supported by this builder" << std::endl; return 2; }
		}
		else if (a == "-h" || a == "--help" || a == "--usage") {
			std::cout << "Usage: bowtie2-build [options]* <reference_in> <bt2_index_base>\n"
			             "    reference_in            comma-separated list of files with ref sequences\n"
			             "    bt2_index_base          write .bt2/.bt2l data to files with this dir/basename\n"
			             "Options: -f, -c, --large-index, -o/--offrate <int>, -t/--ftabchars <int>, --gpu <int>, -q/--quiet, --threads <int> (ignored)\n";
			return 1;
		}
		else if (a == "--version") { std::cout << argv[0] << " (bowtie2_amd index builder, writes bowtie2 v2.5.5-compatible indexes)" << std::endl; return 1; }
		else if (a.size() > 1 && a[0] == '-') { std::cerr << "Error: unknown option " << a << std::endl; return 2; }
		else pos.push_back(a);
	}
	if (pos.size() < 1) { std::cerr << "No input sequence or sequence file specified!" << std::endl; return 2; }
	if (pos.size() < 2) { std::cerr << "No output file specified!" << std::endl; return 2; }
	o.infile = pos[0]; o.outfile = pos[1];
	if (o.line_rate != -1 && o.line_rate != (o.large ? 7 : 6)) { std::cerr << "Error: only the default line rate (6 for .bt2, 7 for .bt2l) is supported: the aligners read no other side size" << std::endl; return 2; }
	if (o.ftab_chars < 1 || o.ftab_chars > 15 || o.off_rate < 0 || o.off_rate > 30) { std::cerr << "Error: --ftabchars must be in [1,15] and --offrate in [0,30]" << std::endl; return 2; }
	return 0;
}

// input -> RefInput (files: plain or .gz FASTA, comma separated; -c: sequences named 0,1,2,... as the reference does)
inline bool read_build_input(const CliOpts& o, RefInput& in, std::string& err) {
	std::vector<std::string> items;
	{ std::stringstream ss(o.infile); std::string t; while (std::getline(ss, t, ',')) if (!t.empty()) items.push_back(t); }
	if (items.empty()) { err = "Tokenized input file list was empty!"; return false; }
	uint64_t seqs = 0;
	if (o.cmdline_seqs) {
		std::string all;
		for (size_t i = 0; i < items.size(); i++) all += ">" + std::to_string(i) + "\n" + items[i] + "\n";
		MemSource src(all.data(), all.size());
		return scan_fasta(src, in, seqs, err);
	}
	bool any = false;
	for (const std::string& p : items) {
		GzSource src(p);
		if (!src.ok()) { err = "Error: could not open " + p; return false; }
		if (src.at_end()) {
			if (src.io_error()) { err = "Error: reading " + p + " failed (corrupt or truncated compressed file?); no index written"; return false; }
			std::cerr << "Warning: Empty fasta file: '" << p << "'" << std::endl; continue;
		}
		any = true;
		if (!scan_fasta(src, in, seqs, err)) return false;
		if (src.io_error()) { err = "Error: reading " + p + " failed (corrupt or truncated compressed file?); no index written"; return false; }
	}
	if (!any) { err = "Warning: All fasta inputs were empty"; return false; }
	return true;
}

template <class Bk>
int build_main(int argc, const char** argv, bool large_default) {
	CliOpts o;
	const int pr = parse_build_args(argc, argv, large_default, o);
	if (pr) return pr == 1 ? 0 : 1;
	Params P;
	P.off_size = o.large ? 8 : 4; P.line_rate = o.large ? 7 : 6; P.off_rate = o.off_rate; P.ftab_chars = o.ftab_chars; P.write_ref = !o.no_ref || o.just_ref; P.just_ref = o.just_ref;
	std::string err;
	if (!Bk::init(o.gpu, err)) { std::cerr << err << std::endl; return 1; }
	RefInput in;
	const double t0 = wall_now();
	if (!read_build_input(o, in, err)) { std::cerr << err << std::endl; return 1; }
	BuildStats st;
	st.t_parse = wall_now() - t0;
	if (!build_index_files<Bk>(in, o.outfile, P, st, err, &wall_now)) { std::cerr << err << std::endl; return 1; }
	if (!o.quiet)
		std::cerr << "bowtie2-build (" << Bk::name() << "): " << st.len << " bases, " << st.n_pat << " sequences, " << st.n_frag << " fragments; parse " << st.t_parse
		          << " s, forward index " << st.t_fw << " s (" << st.tied_fw << " suffixes tied after the first sort, " << st.rounds_fw << " doubling rounds), mirror index "
		          << st.t_bw << " s (" << st.rounds_bw << " rounds), file output " << st.t_write << " s" << std::endl;
	return 0;
}

} } // namespace bt2g::build
#endif
