// bt2g_build_fasta.hpp -- host side of the index builder: FASTA -> reference records, names, joined text.
//
// Restates (behaviour, not code) what bowtie2-build does before suffix sorting:
//   fastaRefReadSize / fastaRefReadSizes   ref_read.cpp:28-303    the RefRecord list ("szs") and the .4 base stream
//   fastaRefReadAppend                     ref_read.h:153-312     the joined text and the reference names
//   BitPairReference::szsFromFasta         reference.cpp:587-668  <base>.3 / <base>.4
//   Ebwt::joinToDisk / szsToDisk           bt2_idx.h:2695-2806, bt2_io.cpp:933-959   nPat, plen[], nFrag, rstarts[]
//   reverseRefRecords                      ref_read.cpp:185-240   the records of the mirror index
// A record is one stretch of unambiguous bases: `off` ambiguous characters (N, IUPAC codes, '-') in front of it,
// `len` bases, `first` = it opens a new sequence.  Character classes are asc2dnacat (alphabet.cpp:36-58).
#ifndef BT2G_BUILD_FASTA_HPP_
#define BT2G_BUILD_FASTA_HPP_

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <zlib.h>

namespace bt2g { namespace build {

struct RefRec { uint64_t off, len; bool first; };

struct RefInput {
	std::vector<RefRec> recs;          // in input order, as the reference's `szs`
	std::vector<std::string> names;    // one per sequence (full header line, as the reference stores it)
	std::vector<uint8_t> joined;       // codes 0..3 of every unambiguous base, in input order
	uint64_t both_tot = 0;             // bases + ambiguous characters
};

// 1 = A/C/G/T, 2 = IUPAC ambiguity code incl. N, 3 = '-', 0 = anything else (ignored inside a sequence)
inline int dna_cat(int c) {
	switch (c) {
		case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return 1;
		case 'B': case 'D': case 'H': case 'K': case 'M': case 'N': case 'R': case 'S': case 'V': case 'W': case 'X': case 'Y':
		case 'b': case 'd': case 'h': case 'k': case 'm': case 'n': case 'r': case 's': case 'v': case 'w': case 'x': case 'y': return 2;
		case '-': return 3;
		default: return 0;
	}
}
inline int dna_code(int c) {
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; default: return 3; }
}

// Byte source with the end-of-input behaviour of the reference's FileBuf: get() returns -1 at the end, at_end() is
// already true once the last byte has been handed out.
class ByteSource {
public:
	virtual ~ByteSource() {}
	int get() { if (cur_ == n_ && !fill()) return -1; return buf_[cur_++]; }
	bool at_end() { if (cur_ == n_) fill(); return cur_ == n_; }
	// bulk access for the base-reading loop: the bytes buffered right now (0 at the end of input), and a way to consume them
	size_t window(const uint8_t*& p) { if (cur_ == n_ && !fill()) return 0; p = buf_ + cur_; return n_ - cur_; }
	void advance(size_t k) { cur_ += k; }
	// the input ended because reading failed (corrupt or truncated compressed file), not because it was read to its end: the
	// caller must not build an index of what happened to come through
	virtual bool io_error() const { return false; }
protected:
	virtual size_t read_some(uint8_t* dst, size_t cap) = 0;
private:
	bool fill() { if (done_) return false; n_ = read_some(buf_, sizeof(buf_)); cur_ = 0; if (n_ == 0) done_ = true; return n_ > 0; }
	uint8_t buf_[1 << 16];
	size_t cur_ = 0, n_ = 0;
	bool done_ = false;
};
class GzSource : public ByteSource {       // plain or gzip-compressed file (zlib reads both)
public:
	explicit GzSource(const std::string& p) { f_ = gzopen(p.c_str(), "rb"); if (f_) gzbuffer(f_, 1 << 20); }
	~GzSource() override { if (f_) gzclose(f_); }
	bool ok() const { return f_ != nullptr; }
	bool io_error() const override { return io_error_; }
protected:
	size_t read_some(uint8_t* dst, size_t cap) override {
		const int n = gzread(f_, dst, (unsigned)cap);
		if (n > 0) return (size_t)n;
		// 0 = end of input, but zlib also returns 0 for a gzip stream that stops short (Z_BUF_ERROR); < 0 = corrupt data
		int zerr = Z_OK;
		(void)gzerror(f_, &zerr);
		if (n < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) io_error_ = true;
		return 0;
	}
private:
	gzFile f_ = nullptr;
	bool io_error_ = false;
};
class MemSource : public ByteSource {      // -c sequences and in-memory genomes: ">name\nSEQ\n" per entry
public:
	MemSource(const char* p, size_t n) : p_(p), n_(n) {}
protected:
	size_t read_some(uint8_t* dst, size_t cap) override {
		const size_t k = n_ - pos_ < cap ? n_ - pos_ : cap;
		for (size_t i = 0; i < k; i++) dst[i] = (uint8_t)p_[pos_ + i];
		pos_ += k;
		return k;
	}
private:
	const char* p_; size_t n_, pos_ = 0;
};

// One input stream -> records, names, joined bases appended to `out`.  Returns false on a malformed start.
inline bool scan_fasta(ByteSource& in, RefInput& out, uint64_t& seqs_read, std::string& err) {
	auto isnl = [](int c) { return c == '\n' || c == '\r'; };
	int c;
	// leading whitespace, then the first '>'
	do { c = in.get(); } while (c != -1 && (c == ' ' || (c >= 9 && c <= 13)));
	if (c == -1) { out.recs.push_back(RefRec{0, 0, true}); out.names.push_back(std::to_string(seqs_read)); seqs_read++; return true; }   // "Empty input file": one empty record
	if (c != '>') { err = "Reference file does not seem to be a FASTA file"; return false; }
	int lastc = '>';
	for (;;) {
		uint64_t off = 0, len = 0;
		bool first = true, finished = false;
		std::string name;
		if (lastc == '>') {
			// the name line; a name line directly followed by another one is an empty sequence that leaves no trace
			for (;;) {
				name.clear();
				c = in.get();
				while (c != -1 && !isnl(c)) { name.push_back((char)c); c = in.get(); }
				while (isnl(c)) c = in.get();
				if (c != '>') break;
			}
			if (c == -1) { lastc = -1; finished = true; }     // header at the very end: an empty sequence
		} else {
			first = false;
			off = 1;             // the ambiguous character that ended the previous record
			c = in.get();
			if (c == -1) { lastc = -1; finished = true; }
		}
		if (!finished) {
			// ambiguous characters in front of the stretch
			for (;;) {
				const int cat = dna_cat(c);
				if (cat == 1) break;
				if (cat >= 2) off++;
				else if (c == '>') { lastc = '>'; finished = true; break; }
				c = in.get();
				if (c == -1) { lastc = -1; finished = true; break; }
			}
		}
		if (!finished) {
			// c holds the stretch's first base.  Bulk loop over the buffered bytes: 0-3 = base code, 4 = ignored inside a
			// sequence (whitespace, digits, ...), 5 = ends the stretch ('>' or an ambiguous character)
			static const struct Lut { uint8_t t[256]; Lut() { for (int i = 0; i < 256; i++) { const int cat = dna_cat(i); t[i] = cat == 1 ? (uint8_t)dna_code(i) : (cat >= 2 || i == '>') ? 5 : 4; } } } lut;
			out.joined.push_back((uint8_t)dna_code(c)); len++;
			c = -1;
			for (bool stop = false; !stop; ) {
				const uint8_t* p;
				const size_t n = in.window(p);
				if (n == 0) { c = -1; break; }
				const size_t base = out.joined.size();
				out.joined.resize(base + n);
				uint8_t* dst = out.joined.data() + base;
				size_t k = 0, m = 0;
				for (; k < n; k++) {
					const uint8_t v = lut.t[p[k]];
					if (v < 4) dst[m++] = v;
					else if (v == 5) { c = p[k]; stop = true; k++; break; }
				}
				out.joined.resize(base + m);
				len += m;
				in.advance(k);
			}
			lastc = c;
		}
		out.both_tot += off + len;
		if (first) {
			if (name.empty()) name = std::to_string(seqs_read);
			out.names.push_back(name);
			seqs_read++;
		}
		if (!(len == 0 && off == 0 && !first)) out.recs.push_back(RefRec{off, len, first});
		if (in.at_end()) break;
	}
	return true;
}

// The records of the entirely reversed text (mirror index): reverseRefRecords.
inline void reverse_records(const std::vector<RefRec>& src, std::vector<RefRec>& dst) {
	dst.clear();
	std::vector<RefRec> cur;
	for (int64_t i = (int64_t)src.size() - 1; i >= 0; i--) {
		bool first = (i == (int64_t)src.size() - 1 || src[(size_t)i + 1].first);
		if (src[(size_t)i].len || (first && src[(size_t)i].off == 0)) { cur.push_back(RefRec{0, src[(size_t)i].len, first}); first = false; }
		if (src[(size_t)i].off) cur.push_back(RefRec{src[(size_t)i].off, 0, first});
	}
	for (size_t i = 0; i < cur.size(); i++) {
		if (i + 1 < cur.size() && cur[i].off != 0 && !cur[i + 1].first) { dst.push_back(RefRec{cur[i].off, cur[i + 1].len, cur[i].first}); i++; }
		else dst.push_back(cur[i]);
	}
}

// nPat, plen[], nFrag, rstarts[] as joinToDisk + szsToDisk write them (rstarts from `recs_dir`: the forward records, or
// the reversed ones with pattern ids and offsets inverted for the mirror index)
struct JoinInfo { uint64_t n_pat = 0, n_frag = 0; std::vector<uint64_t> plen, rstarts; };
inline void join_info(const std::vector<RefRec>& fw_recs, const std::vector<RefRec>& recs_dir, bool reverse, JoinInfo& ji) {
	ji = JoinInfo();
	for (const RefRec& r : fw_recs) { if (r.len > 0) ji.n_frag++; if (r.first) { ji.n_pat++; ji.plen.push_back(r.len + r.off); } else ji.plen.back() += r.len + r.off; }
	uint64_t seq = 0, off = 0, totlen = 0;
	for (const RefRec& r : recs_dir) {
		if (r.first) off = 0;
		off += r.off;
		if (r.first) seq++;
		if (r.len == 0) continue;
		uint64_t seqm1 = seq - 1, fwoff = off;
		if (reverse) { seqm1 = ji.n_pat - seqm1 - 1; fwoff = ji.plen[seqm1] - (off + r.len); }
		ji.rstarts.push_back(totlen); ji.rstarts.push_back(seqm1); ji.rstarts.push_back(fwoff);
		totlen += r.len;
		off += r.len;
	}
}

} } // namespace bt2g::build
#endif
