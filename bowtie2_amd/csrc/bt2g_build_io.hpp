// bt2g_build_io.hpp -- the index builder above the suffix sorter: input -> records -> both directions -> files.
//
// Writes <base>.{1,2,3,4,rev.1,rev.2}.bt2[l] byte for byte as bowtie2-build-{s,l} does (format: SURVEY.md Appendix B;
// writers restated: Ebwt::writeFromMemory(justHeader) bt2_io.cpp:801-823, joinToDisk bt2_idx.h:2695-2806, szsToDisk
// bt2_io.cpp:933-959, buildToDisk bt2_idx.h:2829-3174, the name list initFromVector :1194-1199, szsFromFasta
// reference.cpp:587-668).  Templated on the primitive backend like bt2g_build_core.hpp.
#ifndef BT2G_BUILD_IO_HPP_
#define BT2G_BUILD_IO_HPP_

#include "bt2g_build_core.hpp"
#include "bt2g_build_fasta.hpp"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <unistd.h>
#include <vector>

namespace bt2g { namespace build {

class OutFile {
public:
	explicit OutFile(const std::string& p) : path_(p) { f_ = fopen(p.c_str(), "wb"); if (f_) setvbuf(f_, nullptr, _IOFBF, 1 << 22); }
	~OutFile() { if (f_) fclose(f_); }
	bool ok() const { return f_ != nullptr && !bad_; }
	void raw(const void* p, size_t n) { if (f_ && n && fwrite(p, 1, n, f_) != n) bad_ = true; }
	void i32(int32_t v) { raw(&v, 4); }
	void off(uint64_t v, int off_size) { if (off_size == 4) { const uint32_t x = (uint32_t)v; raw(&x, 4); } else raw(&v, 8); }
	void offs(const std::vector<uint64_t>& v, int off_size) {
		if (off_size == 8) { raw(v.data(), v.size() * 8); return; }
		std::vector<uint32_t> t(1 << 16);
		for (size_t i = 0; i < v.size(); i += t.size()) {
			const size_t k = v.size() - i < t.size() ? v.size() - i : t.size();
			for (size_t j = 0; j < k; j++) t[j] = (uint32_t)v[i + j];
			raw(t.data(), k * 4);
		}
	}
	bool close() { if (f_) { if (fclose(f_) != 0) bad_ = true; f_ = nullptr; } return !bad_; }
private:
	std::string path_; FILE* f_ = nullptr; bool bad_ = false;
};

struct BuildStats { uint64_t len = 0, n_pat = 0, n_frag = 0; uint32_t rounds_fw = 0, rounds_bw = 0; uint64_t tied_fw = 0, tied_bw = 0; double t_parse = 0, t_fw = 0, t_bw = 0, t_write = 0; };

inline bool write_ebwt_files(const std::string& p1, const std::string& p2, const Params& P, bool reverse, uint64_t len,
                             const JoinInfo& ji, const EbwtImage& im, const std::vector<std::string>& names, std::string& err) {
	OutFile o1(p1), o2(p2);
	if (!o1.ok() || !o2.ok()) { err = "Could not open index file for writing: \"" + (o1.ok() ? p2 : p1) + "\""; return false; }
	const int os = P.off_size;
	o1.i32(1); o2.i32(1);
	o1.off(len, os); o1.i32(P.line_rate); o1.i32(2); o1.i32(P.off_rate); o1.i32(P.ftab_chars);
	int32_t flags = 1; if (reverse) flags |= 4;          // EBWT_ENTIRE_REV
	o1.i32(-flags);
	o1.off(ji.n_pat, os);
	for (uint64_t v : ji.plen) o1.off(v, os);
	o1.off(ji.n_frag, os);
	o1.offs(ji.rstarts, os);
	o1.raw(im.ebwt.data(), im.ebwt.size());
	o1.off(im.zoff, os);
	for (int i = 0; i < 5; i++) o1.off(im.fchr[i], os);
	o1.offs(im.ftab, os);
	o1.offs(im.eftab, os);
	for (const std::string& nm : names) { o1.raw(nm.data(), nm.size()); o1.raw("\n", 1); }
	o1.raw("\0", 1);
	o2.offs(im.offs, os);
	if (!o1.close() || !o2.close()) { err = "An error occurred writing the index to disk.  Please check if the disk is full."; return false; }
	return true;
}

inline bool write_ref_files(const std::string& p3, const std::string& p4, int off_size, const RefInput& in, std::string& err) {
	OutFile o3(p3), o4(p4);
	if (!o3.ok() || !o4.ok()) { err = "Could not open index file for writing: \"" + (o3.ok() ? p4 : p3) + "\""; return false; }
	o3.i32(1);
	o3.off(in.recs.size(), off_size);
	for (const RefRec& r : in.recs) { o3.off(r.off, off_size); o3.off(r.len, off_size); const uint8_t f = r.first ? 1 : 0; o3.raw(&f, 1); }
	// 4 bases per byte, first base in the low bits (BitpairOutFileBuf, filebuf.h:558-619)
	const uint64_t n = in.joined.size();
	std::vector<uint8_t> buf(1 << 20);
	for (uint64_t b0 = 0; b0 < (n + 3) / 4; b0 += buf.size()) {
		const uint64_t nb = (n + 3) / 4 - b0 < buf.size() ? (n + 3) / 4 - b0 : buf.size();
		for (uint64_t b = 0; b < nb; b++) {
			const uint64_t i = (b0 + b) * 4;
			unsigned v = 0;
			for (int k = 0; k < 4 && i + k < n; k++) v |= (unsigned)in.joined[i + k] << (2 * k);
			buf[b] = (uint8_t)v;
		}
		o4.raw(buf.data(), nb);
	}
	if (!o3.close() || !o4.close()) { err = "An error occurred writing the index to disk.  Please check if the disk is full."; return false; }
	return true;
}

// Everything after the input has been scanned.  `Bk` = primitive backend.
template <class Bk>
bool build_index_files(const RefInput& in, const std::string& out_base, const Params& P, BuildStats& st, std::string& err,
                       double (*now)()) {
	const uint64_t len = in.joined.size();
	if (len == 0) { err = "Error: No unambiguous stretches of characters in the input.  Aborting..."; return false; }
	const uint64_t max_len = P.off_size == 4 ? 0xfffffffeull : 0xfffffffffffffffeull;
	if (len > max_len) { err = "Error: Reference sequence has more than 2^32-1 characters!  Please build a large index instead (bowtie2-build-l)."; return false; }
	const std::string ext = P.off_size == 4 ? "bt2" : "bt2l";
	st.len = len;
	// Every file is written under a temporary name and gets its real name only when the whole build has succeeded: a build that fails
	// half way (device error, out of memory, full disk) leaves no index files behind, complete-looking or not.
	struct Outputs {
		std::vector<std::pair<std::string, std::string>> files;      // (temporary, final)
		bool committed = false;
		std::string tmp(const std::string& final_name) { files.emplace_back(final_name + ".tmp" + std::to_string((long)getpid()), final_name); return files.back().first; }
		bool commit(std::string& err) {
			// the .1 files are what the aligner probes for: they are moved last, and a rename that fails takes back the ones already moved
			std::stable_sort(files.begin(), files.end(), [](const std::pair<std::string, std::string>& a, const std::pair<std::string, std::string>& b) {
				auto is1 = [](const std::string& n) { return n.find(".1.bt2") != std::string::npos; };
				return !is1(a.second) && is1(b.second);
			});
			for (size_t i = 0; i < files.size(); i++) {
				if (rename(files[i].first.c_str(), files[i].second.c_str()) != 0) {
					err = "Could not move index file into place: \"" + files[i].second + "\"";
					for (size_t k = 0; k < i; k++) (void)remove(files[k].second.c_str());
					return false;
				}
			}
			committed = true;
			return true;
		}
		~Outputs() { if (!committed) for (auto& f : files) { (void)remove(f.first.c_str()); } }
	} outs;
	double t0 = now();
	if (P.write_ref) {
		const std::string t3 = outs.tmp(out_base + ".3." + ext), t4 = outs.tmp(out_base + ".4." + ext);
		if (!write_ref_files(t3, t4, P.off_size, in, err)) return false;
	}
	st.t_write += now() - t0;
	if (P.just_ref) return outs.commit(err);
	std::vector<RefRec> rrecs;
	reverse_records(in.recs, rrecs);
	JoinInfo ji_fw, ji_bw;
	join_info(in.recs, in.recs, false, ji_fw);
	join_info(in.recs, rrecs, true, ji_bw);
	st.n_pat = ji_fw.n_pat; st.n_frag = ji_fw.n_frag;
	auto run = [&](auto& builder) -> bool {
		if (!builder.upload_text(in.joined.data(), len)) { err = builder.err; return false; }
		for (int dir = 0; dir < 2; dir++) {
			EbwtImage im;
			t0 = now();
			if (!builder.build(dir == 1, im)) { err = builder.err; builder.release_text(); return false; }
			// a primitive of the backend failed somewhere on the way (the builder's own checks cannot see a kernel or copy that did nothing)
			if (!Bk::error().empty()) { err = "index build failed: " + Bk::error(); builder.release_text(); return false; }
			(dir ? st.t_bw : st.t_fw) = now() - t0;
			(dir ? st.rounds_bw : st.rounds_fw) = im.rounds;
			(dir ? st.tied_bw : st.tied_fw) = im.tied_after_first;
			t0 = now();
			const std::string b = dir ? out_base + ".rev" : out_base;
			const std::string t1 = outs.tmp(b + ".1." + ext), t2 = outs.tmp(b + ".2." + ext);
			if (!write_ebwt_files(t1, t2, P, dir == 1, len, dir ? ji_bw : ji_fw, im, in.names, err)) { builder.release_text(); return false; }
			st.t_write += now() - t0;
		}
		builder.release_text();
		return outs.commit(err);
	};
	// 32-bit text positions while they fit (BT2G_BUILD_FORCE_IDX64: test hook for the 64-bit path on small inputs)
	if (len < 0xfffffffeull && !getenv("BT2G_BUILD_FORCE_IDX64")) { Builder<Bk, uint32_t> b(P); return run(b); }
	Builder<Bk, uint64_t> b(P);
	return run(b);
}

} } // namespace bt2g::build
#endif
