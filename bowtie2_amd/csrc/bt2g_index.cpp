// bt2g_index.cpp -- see bt2g_index.hpp.
#include "bt2g_index.hpp"
#include "../../include/bt2g.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <sys/stat.h>

namespace bt2g {

namespace {

struct File {
	FILE* f = nullptr;
	explicit File(const std::string& p) { f = fopen(p.c_str(), "rb"); }
	~File() { if (f) fclose(f); }
	bool ok() const { return f != nullptr; }
	bool read(void* dst, size_t n) { return n == 0 || fread(dst, 1, n, f) == n; }
	bool skip(uint64_t n) { return fseeko(f, (off_t)n, SEEK_CUR) == 0; }
	bool off(int off_size, uint64_t& v) {
		if (off_size == 4) { uint32_t x; if (!read(&x, 4)) return false; v = x; return true; }
		return read(&v, 8);
	}
	bool raw(std::vector<uint8_t>& v, uint64_t nbytes) {
		if (nbytes > remaining()) return false;      // a corrupt header must not turn into a huge allocation
		v.resize(nbytes);
		return read(v.data(), nbytes);
	}
	// a large section: read into `v`, or (lazy) left in the file and described by `sp`
	bool raw_or_span(std::vector<uint8_t>& v, FileSpan& sp, const std::string& path, uint64_t nbytes, bool lazy) {
		if (!lazy) return raw(v, nbytes);
		if (nbytes > remaining()) return false;
		const off_t cur = ftello(f);
		if (cur < 0) return false;
		sp.path = path; sp.off = (uint64_t)cur; sp.nbytes = nbytes;
		return skip(nbytes);
	}
	uint64_t remaining() {
		const off_t cur = ftello(f);
		struct stat st;
		if (cur < 0 || fstat(fileno(f), &st) != 0 || st.st_size < cur) return 0;
		return (uint64_t)(st.st_size - cur);
	}
};

bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

int load_ebwt(const std::string& p1, const std::string& p2, int off_size, bool fw, HostEbwt& e, std::string& err, bool lazy) {
	File f(p1);
	if (!f.ok()) { err = "cannot open " + p1; return BT2G_ERR_IO; }
	int32_t one = 0, lines_per_side = 0;
	if (!f.read(&one, 4) || one != 1) { err = p1 + ": bad endianness sentinel"; return BT2G_ERR_FORMAT; }
	bool ok = f.off(off_size, e.len) && f.read(&e.line_rate, 4) && f.read(&lines_per_side, 4) &&
	          f.read(&e.off_rate, 4) && f.read(&e.ftab_chars, 4) && f.read(&e.flags, 4);
	if (!ok) { err = p1 + ": truncated header"; return BT2G_ERR_FORMAT; }
	if (e.line_rate < 5 || e.line_rate > 12 || e.ftab_chars < 1 || e.ftab_chars > 15 || e.off_rate < 0 || e.off_rate > 30) {
		err = p1 + ": implausible header"; return BT2G_ERR_FORMAT;
	}
	// The resident suffix array keeps, per row, the LF steps the reference's getOffset would have walked in 16 bits (joff_pack,
	// bt2g_device.hpp).  The sample is by row, so a walk's length is geometric with mean 2^offRate: above 15 most rows would not fit; for
	// the rates below that, bt2g_index_load counts the rows that do not while it builds the array and refuses the index if there are any.
	if (e.off_rate > 15) { err = p1 + ": --offrate above 15 is not supported by this build (suffix-array sample too sparse)"; return BT2G_ERR_UNSUPPORTED; }
	// Colorspace indexes (flags & 2) and pre-2.0 "each stretch reversed" mirrors are not supported.
	if (e.flags < 0 && ((-e.flags) & 2)) { err = p1 + ": colorspace index"; return BT2G_ERR_UNSUPPORTED; }
	if (!fw && !(e.flags < 0 && ((-e.flags) & 4))) { err = p1 + ": mirror index is not an entire-reverse index (built by bowtie2 < 2.0?)"; return BT2G_ERR_UNSUPPORTED; }
	e.side_sz = 1u << e.line_rate;
	e.side_bwt_sz = e.side_sz - 4u * (uint32_t)off_size;
	e.side_bwt_len = e.side_bwt_sz * 4u;
	if (e.side_bwt_len != 48u * (uint32_t)off_size) { err = p1 + ": unsupported line rate"; return BT2G_ERR_UNSUPPORTED; }
	const uint64_t bwt_sz = e.len / 4 + 1;
	e.num_sides = (bwt_sz + e.side_bwt_sz - 1) / e.side_bwt_sz;
	e.ebwt_tot_len = e.num_sides * e.side_sz;
	e.ftab_len = (1ull << (2 * e.ftab_chars)) + 1;
	e.eftab_len = 2ull * (uint64_t)e.ftab_chars;
	e.offs_len = (e.len + 1 + (1ull << e.off_rate) - 1) >> e.off_rate;
	if (!f.off(off_size, e.n_pat) || e.n_pat > f.remaining() / (uint64_t)off_size || !f.raw(e.plen, e.n_pat * off_size)) { err = p1 + ": truncated plen"; return BT2G_ERR_FORMAT; }
	if (!f.off(off_size, e.n_frag) || e.n_frag > f.remaining() / (3ull * (uint64_t)off_size)) { err = p1 + ": truncated"; return BT2G_ERR_FORMAT; }
	if (fw) {
		if (!f.raw(e.rstarts, e.n_frag * 3 * off_size)) { err = p1 + ": truncated rstarts"; return BT2G_ERR_FORMAT; }
	} else if (!f.skip(e.n_frag * 3 * off_size)) { err = p1 + ": truncated"; return BT2G_ERR_FORMAT; }
	if (!f.raw_or_span(e.ebwt, e.ebwt_span, p1, e.ebwt_tot_len, lazy)) { err = p1 + ": truncated ebwt"; return BT2G_ERR_FORMAT; }
	if (!f.off(off_size, e.zoff)) { err = p1 + ": truncated"; return BT2G_ERR_FORMAT; }
	for (int i = 0; i < 5; i++) if (!f.off(off_size, e.fchr[i])) { err = p1 + ": truncated fchr"; return BT2G_ERR_FORMAT; }
	if (!f.raw(e.ftab, e.ftab_len * off_size) || !f.raw(e.eftab, e.eftab_len * off_size)) {
		err = p1 + ": truncated ftab"; return BT2G_ERR_FORMAT;
	}
	if (fw) {
		// reference names: '\n'-separated, '\0'-terminated
		std::string cur; bool any = false;
		for (;;) {
			int c = fgetc(f.f);
			if (c == EOF || c == 0) { if (any) e.refnames.push_back(cur); break; }
			if (c == '\n') { e.refnames.push_back(cur); cur.clear(); any = false; }
			else { cur.push_back((char)c); any = true; }
		}
		File g(p2);
		if (!g.ok()) { err = "cannot open " + p2; return BT2G_ERR_IO; }
		if (!g.read(&one, 4) || one != 1 || !g.raw_or_span(e.offs, e.offs_span, p2, e.offs_len * off_size, lazy)) {
			err = p2 + ": truncated SA sample"; return BT2G_ERR_FORMAT;
		}
	}
	return 0;
}

int load_ref(const std::string& p3, const std::string& p4, int off_size, HostRef& r, std::string& err, bool lazy) {
	File f(p3);
	if (!f.ok()) { err = "cannot open " + p3; return BT2G_ERR_IO; }
	int32_t one = 0;
	if (!f.read(&one, 4) || one != 1 || !f.off(off_size, r.nrecs) || r.nrecs == 0) {
		err = p3 + ": bad header"; return BT2G_ERR_FORMAT;
	}
	if (r.nrecs > f.remaining() / (2ull * (uint64_t)off_size + 1)) { err = p3 + ": truncated"; return BT2G_ERR_FORMAT; }
	r.rec_refpos.resize(r.nrecs); r.rec_bufpos.resize(r.nrecs); r.rec_len.resize(r.nrecs);
	uint64_t cumsz = 0, cumlen = 0;
	for (uint64_t i = 0; i < r.nrecs; i++) {
		uint64_t off, len; uint8_t first;
		if (!f.off(off_size, off) || !f.off(off_size, len) || !f.read(&first, 1)) { err = p3 + ": truncated"; return BT2G_ERR_FORMAT; }
		if (first) {
			if (r.nrefs > 0) r.ref_lens.push_back(cumlen);
			r.ref_rec_offs.push_back(i);
			cumlen = 0;
			r.nrefs++;
		} else if (i == 0) { err = p3 + ": first record not marked first"; return BT2G_ERR_FORMAT; }
		r.rec_refpos[i] = cumlen + off;   // stretch starts after `off` ambiguous characters
		r.rec_bufpos[i] = cumsz;
		r.rec_len[i] = len;
		cumsz += len;
		cumlen += off + len;
	}
	r.ref_lens.push_back(cumlen);
	r.ref_rec_offs.push_back(r.nrecs);
	r.buf_sz = cumsz;
	File g(p4);
	if (!g.ok()) { err = "cannot open " + p4; return BT2G_ERR_IO; }
	if (!g.raw_or_span(r.buf, r.buf_span, p4, (cumsz + 3) / 4, lazy)) { err = p4 + ": truncated"; return BT2G_ERR_FORMAT; }
	if (!lazy) r.buf.resize(r.buf.size() + 16, 0); // slack so device code may read a few bytes past the end
	return 0;
}

} // namespace

uint64_t HostIndex::plen_at(uint64_t i) const {
	if (off_size == 4) { uint32_t v; memcpy(&v, fw.plen.data() + i * 4, 4); return v; }
	uint64_t v; memcpy(&v, fw.plen.data() + i * 8, 8); return v;
}

static int load_index_impl(const std::string& base, HostIndex& out, std::string& err, bool lazy);

int load_index(const std::string& base, HostIndex& out, std::string& err, bool lazy) {
	// no exception may cross the C ABI (bt2g_index_load): allocation failures and length errors become status codes
	try { return load_index_impl(base, out, err, lazy); }
	catch (const std::bad_alloc&) { err = "out of host memory while reading the index"; return BT2G_ERR_NOMEM; }
	catch (const std::exception& e) { err = std::string("malformed index: ") + e.what(); return BT2G_ERR_FORMAT; }
}

static int load_index_impl(const std::string& base, HostIndex& out, std::string& err, bool lazy) {
	std::string ext = "bt2";
	out.off_size = 4;
	if (!exists(base + ".1.bt2")) {
		if (!exists(base + ".1.bt2l")) { err = "no index found at " + base + ".1.bt2[l]"; return BT2G_ERR_IO; }
		ext = "bt2l"; out.off_size = 8;
	}
	int rc = load_ebwt(base + ".1." + ext, base + ".2." + ext, out.off_size, true, out.fw, err, lazy);
	if (rc) return rc;
	rc = load_ebwt(base + ".rev.1." + ext, "", out.off_size, false, out.bw, err, lazy);
	if (rc) return rc;
	if (out.bw.len != out.fw.len || out.bw.ftab_chars != out.fw.ftab_chars) { err = "forward/mirror index mismatch"; return BT2G_ERR_FORMAT; }
	rc = load_ref(base + ".3." + ext, base + ".4." + ext, out.off_size, out.ref, err, lazy);
	if (rc) return rc;
	if (out.ref.nrefs != out.fw.n_pat) {
		// The .3 file may list empty (all-N) references the .1 file drops; the reference tolerates
		// this (reference.cpp:130-160) -- we only need ids consistent with rstarts, which use .1 ids.
	}
	return 0;
}

} // namespace bt2g
