// bt2g_build_core.hpp -- suffix sorting and FM-index emission for the index builder.
//
// What it replaces: KarkkainenBlockwiseSA (blockwise_sa.h) + Ebwt::buildToDisk (bt2_idx.h:2829-3174).  The reference
// walks the suffix array one row at a time on the host; this is a data-parallel restatement whose output is the same
// bytes: the 2-bit BWT cut into sides with their occ[] tallies, zOff, fchr, ftab/eftab and the SA sample.
//
// Method (MI355X-shaped: everything is a radix sort, a scan or a gather over arrays that sit in HBM):
//   1. the text is packed 2 bits per base; every suffix gets a 64-bit key = its first 29 bases (58 bits) + a 6-bit
//      "characters missing" field, so that the end-of-text sentinel -- which bowtie2 orders AFTER every base
//      (the empty suffix is the last row, bt2_util.cpp:148) -- compares correctly;
//   2. one radix sort of (key, position) pairs; suffixes whose key is unique are final;
//   3. prefix doubling over the still-tied suffixes only: key2 = rank of the suffix h further on, sort the tied ones by
//      (group, key2), split groups, h *= 2, until no ties remain;
//   4. BWT bytes, per-side tallies (exclusive scan), SA sample and the 10-mer table are gathers / histograms.
// `Bk` supplies the primitives (device: rocPRIM sort/scan/select + a grid-stride parallel-for; the host twin used by the
// CPU tests: std:: algorithms).  TIdx is the width of a text position (uint32_t while the text has < 2^32 - 1 bases).
#ifndef BT2G_BUILD_CORE_HPP_
#define BT2G_BUILD_CORE_HPP_

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#ifndef BT2_HD
#if defined(__HIPCC__)
#define BT2_HD __host__ __device__ __forceinline__
#else
#define BT2_HD inline
#endif
#endif

namespace bt2g { namespace build {

constexpr int kKeyBases = 29;          // bases in a first-round key

struct Params {
	int off_size = 4;                  // 4 = .bt2, 8 = .bt2l
	int line_rate = 6;                 // 6 / 7
	int off_rate = 4;
	int ftab_chars = 10;
	bool write_ref = true;             // <base>.3 / <base>.4 (cleared by -r/--noref)
	bool just_ref = false;             // -3/--justref: only those two
};

// One direction of the index, as it goes to disk (host memory)
struct EbwtImage {
	std::vector<uint8_t> ebwt;         // numSides * sideSz
	uint64_t zoff = 0;
	uint64_t fchr[5] = {0, 0, 0, 0, 0};
	std::vector<uint64_t> ftab, eftab; // widened; written at off_size
	std::vector<uint64_t> offs;        // SA sample, widened
	uint64_t num_sides = 0;
	uint32_t rounds = 0;               // doubling rounds it took (diagnostic)
	uint64_t tied_after_first = 0;     // suffixes still tied after the first sort (diagnostic)
};

// packed text: word w holds bases [32w, 32w+32), base j of the word in bits 63-2j .. 62-2j
BT2_HD int text_base(const uint64_t* W, uint64_t i) { return (int)((W[i >> 5] >> (62 - 2 * (i & 31))) & 3); }
// the 32 bases starting at i (bases past the end of the text read as 3: the padding words are all ones)
BT2_HD uint64_t text_window(const uint64_t* W, uint64_t i) {
	const uint64_t w = i >> 5; const unsigned o = (unsigned)(i & 31) * 2;
	return o ? (W[w] << o) | (W[w + 1] >> (64 - o)) : W[w];
}

template <typename T> struct DBuf { T* cur; T* alt; };

template <class Bk, typename TIdx>
struct Builder {
	const Params P;
	uint64_t n = 0;                    // text length
	uint8_t* d_text = nullptr;         // codes, 1 byte per base (device)
	uint64_t* d_W = nullptr;           // packed text (+ 2 padding words)
	std::string err;

	explicit Builder(const Params& p) : P(p) {}

	// text: n codes 0..3 in host memory
	bool upload_text(const uint8_t* text, uint64_t n_) {
		n = n_;
		d_text = Bk::template alloc<uint8_t>(n + 64);
		if (!d_text) { err = "out of device memory (text)"; return false; }
		Bk::upload(d_text, text, n);
		return true;
	}
	void release_text() { if (d_text) Bk::release(d_text); d_text = nullptr; }

	// Build one direction.  reverse: index the reversed text (mirror index).
	bool build(bool reverse, EbwtImage& out) {
		const uint64_t n_ = n;
		const uint64_t nw = (n_ + 31) / 32 + 2;
		d_W = Bk::template alloc<uint64_t>(nw);
		if (!d_W) { err = "out of device memory (packed text)"; return false; }
		{
			const uint8_t* T = d_text; uint64_t* W = d_W;
			Bk::pfor(nw, [=] BT2_HD_LAMBDA (uint64_t w) {
				uint64_t v = 0;
				for (int j = 0; j < 32; j++) {
					const uint64_t i = w * 32 + (uint64_t)j;
					const uint64_t c = i < n_ ? (uint64_t)T[reverse ? n_ - 1 - i : i] : 3ull;
					v = (v << 2) | c;
				}
				W[w] = v;
			});
		}
		const bool ok = sort_and_emit(out);
		Bk::release(d_W); d_W = nullptr;
		return ok;
	}

private:
	bool sort_and_emit(EbwtImage& out) {
		const uint64_t n_ = n;
		const uint64_t* W = d_W;
		// ---- 1+2: first-round keys, one sort ----
		DBuf<uint64_t> key{Bk::template alloc<uint64_t>(n_ + 1), Bk::template alloc<uint64_t>(n_ + 1)};
		DBuf<TIdx> sa{Bk::template alloc<TIdx>(n_ + 1), Bk::template alloc<TIdx>(n_ + 1)};
		TIdx* isa = Bk::template alloc<TIdx>(n_ + 2);
		uint8_t* flag = Bk::template alloc<uint8_t>(n_ + 1);
		auto free_first = [&]() { Bk::release(key.cur); Bk::release(key.alt); Bk::release(sa.alt); Bk::release(flag); key.cur = key.alt = nullptr; sa.alt = nullptr; flag = nullptr; };
		if (!key.cur || !key.alt || !sa.cur || !sa.alt || !isa || !flag) { free_first(); Bk::release(sa.cur); Bk::release(isa); err = "out of device memory (suffix sort)"; return false; }
		{
			uint64_t* k = key.cur; TIdx* v = sa.cur;
			Bk::pfor(n_, [=] BT2_HD_LAMBDA (uint64_t i) {
				const uint64_t miss = i + (uint64_t)kKeyBases > n_ ? i + (uint64_t)kKeyBases - n_ : 0;
				k[i] = (text_window(W, i) & ~0x3full) | miss;      // top 58 bits = 29 bases
				v[i] = (TIdx)i;
			});
		}
		Bk::sort_pairs(key, sa, n_, 0, 64);
		// group heads; rank of a group = row of its head; ISA; which rows are still tied
		uint64_t m = 0;
		TIdx* pos = nullptr;
		{
			const uint64_t* k = key.cur; TIdx* grp = reinterpret_cast<TIdx*>(key.alt);    // the spare key buffer holds the ranks
			Bk::pfor(n_, [=] BT2_HD_LAMBDA (uint64_t r) { grp[r] = (r == 0 || k[r] != k[r - 1]) ? (TIdx)r : (TIdx)0; });
			Bk::inclusive_max(grp, n_);
			const TIdx* s = sa.cur; TIdx* I = isa; uint8_t* f = flag;
			Bk::pfor(n_, [=] BT2_HD_LAMBDA (uint64_t r) {
				I[s[r]] = grp[r];
				const bool head = grp[r] == (TIdx)r, next_head = r + 1 == n_ || grp[r + 1] == (TIdx)(r + 1);
				f[r] = (head && next_head) ? 0 : 1;
			});
			Bk::pfor(1, [=] BT2_HD_LAMBDA (uint64_t) { I[n_] = (TIdx)n_; });     // the empty suffix: last row
			TIdx* pos_full = reinterpret_cast<TIdx*>(key.cur);                    // the sorted keys are no longer needed
			m = Bk::select_index(flag, pos_full, n_);
			if (m > 0) {
				pos = Bk::template alloc<TIdx>(m);
				if (!pos) { free_first(); Bk::release(sa.cur); Bk::release(isa); err = "out of device memory (suffix sort)"; return false; }
				Bk::copy(pos, pos_full, m * sizeof(TIdx));
			}
		}
		free_first();
		out.tied_after_first = m;
		// ---- 3: prefix doubling over the tied suffixes ----
		uint32_t rounds = 0;
		if (m > 0) {
			const uint64_t m0 = m;
			DBuf<uint64_t> ka{Bk::template alloc<uint64_t>(m0), Bk::template alloc<uint64_t>(m0)};
			DBuf<TIdx> ix{Bk::template alloc<TIdx>(m0), Bk::template alloc<TIdx>(m0)};
			TIdx* hd = Bk::template alloc<TIdx>(m0);
			uint8_t* f = Bk::template alloc<uint8_t>(m0);
			DBuf<TIdx> pb{pos, Bk::template alloc<TIdx>(m0)};
			auto free_rounds = [&]() { Bk::release(ka.cur); Bk::release(ka.alt); Bk::release(ix.cur); Bk::release(ix.alt); Bk::release(hd); Bk::release(f); Bk::release(pb.cur); Bk::release(pb.alt); };
			if (!ka.cur || !ka.alt || !ix.cur || !ix.alt || !hd || !f || !pb.alt) { free_rounds(); Bk::release(sa.cur); Bk::release(isa); err = "out of device memory (doubling rounds)"; return false; }
			unsigned idx_bits = 1; while (idx_bits < 64 && (n_ >> idx_bits) != 0) idx_bits++;
			uint64_t h = kKeyBases;
			TIdx* const SA = sa.cur; TIdx* const I = isa;
			while (m > 0) {
				if (++rounds > 64) { free_rounds(); Bk::release(sa.cur); Bk::release(isa); err = "suffix sort did not converge"; return false; }
				const TIdx* p = pb.cur;
				const uint64_t* kk; const uint64_t* kk2 = nullptr;
				if (sizeof(TIdx) == 4) {
					// one composite key: (group << 32) | rank of the suffix h further on
					uint64_t* ck = ka.cur; TIdx* xi = ix.cur;
					Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) {
						const uint64_t i = (uint64_t)SA[p[j]];
						ck[j] = ((uint64_t)I[i] << 32) | (uint64_t)I[i + h];
						xi[j] = (TIdx)i;
					});
					Bk::sort_pairs(ka, ix, m, 0, 32 + (int)idx_bits);
					kk = ka.cur;
				} else {
					// 64-bit positions: two stable passes, least significant key (the rank h further on) first
					uint64_t* k2 = ka.cur; TIdx* xi = ix.cur;
					Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) { const uint64_t i = (uint64_t)SA[p[j]]; k2[j] = (uint64_t)I[i + h]; xi[j] = (TIdx)i; });
					Bk::sort_pairs(ka, ix, m, 0, (int)idx_bits);
					uint64_t* g = ka.cur; const TIdx* xs = ix.cur;
					Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) { g[j] = (uint64_t)I[xs[j]]; });
					Bk::sort_pairs(ka, ix, m, 0, (int)idx_bits);
					uint64_t* k2s = ka.alt; const TIdx* xs2 = ix.cur;
					Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) { k2s[j] = (uint64_t)I[(uint64_t)xs2[j] + h]; });
					kk = ka.cur; kk2 = ka.alt;
				}
				// split the groups, write SA and ISA, keep what is still tied
				const TIdx* xi = ix.cur;
				Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) { hd[j] = (j == 0 || kk[j] != kk[j - 1] || (kk2 && kk2[j] != kk2[j - 1])) ? (TIdx)j : (TIdx)0; });
				Bk::inclusive_max(hd, m);
				const uint64_t mm = m;
				Bk::pfor(m, [=] BT2_HD_LAMBDA (uint64_t j) {
					const TIdx i = xi[j];
					SA[p[j]] = i;
					I[i] = p[hd[j]];
					const bool head = hd[j] == (TIdx)j, next_head = j + 1 == mm || hd[j + 1] == (TIdx)(j + 1);
					f[j] = (head && next_head) ? 0 : 1;
				});
				m = Bk::select(p, f, pb.alt, m);
				TIdx* t = pb.cur; pb.cur = pb.alt; pb.alt = t;
				h *= 2;
			}
			free_rounds();
		}
		out.rounds = rounds;
		// ---- 4: emission ----
		const bool ok = emit(sa.cur, isa, out);
		Bk::release(sa.cur); Bk::release(isa);
		return ok;
	}

	// BWT sides, tallies, SA sample, ftab/eftab, fchr, zOff (Ebwt::buildToDisk)
	bool emit(const TIdx* SA, const TIdx* I, EbwtImage& out) {
		const uint64_t n_ = n;
		const uint64_t* W = d_W;
		const uint32_t side_sz = 1u << P.line_rate, side_bwt_sz = side_sz - 4u * (uint32_t)P.off_size, side_bwt_len = side_bwt_sz * 4u;
		const uint64_t bwt_sz = n_ / 4 + 1;
		const uint64_t num_sides = (bwt_sz + side_bwt_sz - 1) / side_bwt_sz;
		const uint64_t tot = num_sides * side_sz;
		out.num_sides = num_sides;
		uint8_t* d_ebwt = Bk::template alloc<uint8_t>(tot);
		uint64_t* cnt = Bk::template alloc<uint64_t>(4 * num_sides + 4);
		const uint64_t offs_len = (n_ + 1 + (1ull << P.off_rate) - 1) >> P.off_rate;
		uint64_t* d_offs = Bk::template alloc<uint64_t>(offs_len);
		const int fc = P.ftab_chars;
		const uint64_t ftab_len = (1ull << (2 * fc)) + 1;
		uint64_t* d_hist = Bk::template alloc<uint64_t>(ftab_len);
		uint64_t* d_small = Bk::template alloc<uint64_t>(64);
		auto free_all = [&]() { Bk::release(d_ebwt); Bk::release(cnt); Bk::release(d_offs); Bk::release(d_hist); Bk::release(d_small); };
		if (!d_ebwt || !cnt || !d_offs || !d_hist || !d_small) { free_all(); err = "out of device memory (emission)"; return false; }
		// zOff = row of suffix 0
		uint64_t zoff = 0;
		{ Bk::pfor(1, [=] BT2_HD_LAMBDA (uint64_t) { d_small[0] = (uint64_t)I[0]; }); Bk::download(&zoff, d_small, 8); }
		if (n_ == 0) zoff = 0;
		out.zoff = zoff;
		// BWT bytes: one byte = 4 rows, base of row r at bits 2*(r&3); rows past n are padding ('A'); row zOff is stored as 'A'
		const uint64_t nbytes = num_sides * side_bwt_sz;
		Bk::pfor(nbytes, [=] BT2_HD_LAMBDA (uint64_t b) {
			const uint64_t s = b / side_bwt_sz, bo = b % side_bwt_sz;
			unsigned v = 0;
			for (int k = 0; k < 4; k++) {
				const uint64_t r = b * 4 + (uint64_t)k;
				int c = 0;
				if (r <= n_) {
					const uint64_t i = r < n_ ? (uint64_t)SA[r] : n_;
					c = i ? text_base(W, i - 1) : 0;
				}
				v |= (unsigned)c << (2 * k);
			}
			d_ebwt[s * side_sz + bo] = (uint8_t)v;
		});
		// per-side tallies of the 4 characters (the zOff row is not counted, padding is)
		Bk::pfor(num_sides, [=] BT2_HD_LAMBDA (uint64_t s) {
			uint64_t c[4] = {0, 0, 0, 0};
			const uint8_t* p = d_ebwt + s * side_sz;
			for (uint32_t b = 0; b < side_bwt_sz; b++) { const unsigned v = p[b]; c[v & 3]++; c[(v >> 2) & 3]++; c[(v >> 4) & 3]++; c[(v >> 6) & 3]++; }
			if (zoff / side_bwt_len == s) c[0]--;
			for (int k = 0; k < 4; k++) cnt[(uint64_t)k * num_sides + s] = c[k];
		});
		for (int k = 0; k < 4; k++) Bk::exclusive_sum(cnt + (uint64_t)k * num_sides, num_sides);
		{
			const int osz = P.off_size;
			Bk::pfor(num_sides, [=] BT2_HD_LAMBDA (uint64_t s) {
				uint8_t* p = d_ebwt + s * side_sz + side_bwt_sz;
				for (int k = 0; k < 4; k++) {
					const uint64_t v = cnt[(uint64_t)k * num_sides + s];
					for (int q = 0; q < osz; q++) p[k * osz + q] = (uint8_t)(v >> (8 * q));
				}
			});
		}
		// SA sample
		{
			const int orate = P.off_rate;
			Bk::pfor(offs_len, [=] BT2_HD_LAMBDA (uint64_t k) { const uint64_t r = k << orate; d_offs[k] = r < n_ ? (uint64_t)SA[r] : n_; });
		}
		// ftab counts: histogram of the text's fc-mers (every suffix with at least fc characters), slot sufInt+1
		Bk::fill0(d_hist, ftab_len);
		if (n_ >= (uint64_t)fc) {
			const uint64_t cnt_mers = n_ - (uint64_t)fc + 1;
			Bk::pfor(cnt_mers, [=] BT2_HD_LAMBDA (uint64_t i) {
				const uint64_t suf = text_window(W, i) >> (64 - 2 * fc);
				Bk::atomic_add(&d_hist[suf + 1], 1ull);
			});
		}
		out.ebwt.resize(tot);
		Bk::download(out.ebwt.data(), d_ebwt, tot);
		out.offs.resize(offs_len);
		Bk::download(out.offs.data(), d_offs, offs_len * 8);
		std::vector<uint64_t> hist(ftab_len);
		Bk::download(hist.data(), d_hist, ftab_len * 8);
		// the (at most fc) suffixes shorter than fc characters, incl. the empty one: their rows, and for each maximal run of
		// them the fc-mer of the row that follows ("absorbed" into that ftab transition, bt2_idx.h:2984-3001)
		std::vector<uint8_t> absorb(ftab_len, 0);
		{
			const uint64_t nshort = n_ < (uint64_t)fc ? n_ + 1 : (uint64_t)fc;    // suffix lengths 0 .. nshort-1
			std::vector<uint64_t> rows(nshort);
			Bk::pfor(nshort, [=] BT2_HD_LAMBDA (uint64_t k) { d_small[k] = (uint64_t)I[n_ - k]; });       // suffix starting at n-k has k characters
			Bk::download(rows.data(), d_small, nshort * 8);
			for (size_t a = 1; a < rows.size(); a++) { const uint64_t v = rows[a]; size_t b = a; while (b > 0 && rows[b - 1] > v) { rows[b] = rows[b - 1]; b--; } rows[b] = v; }
			size_t a = 0;
			while (a < rows.size()) {
				size_t b = a;
				while (b + 1 < rows.size() && rows[b + 1] == rows[b] + 1) b++;
				const uint64_t cntrun = (uint64_t)(b - a + 1), nxt = rows[b] + 1;
				if (nxt > n_) absorb[ftab_len - 1] = (uint8_t)cntrun;
				else {
					Bk::pfor(1, [=] BT2_HD_LAMBDA (uint64_t) { const uint64_t i = (uint64_t)SA[nxt]; d_small[32] = text_window(W, i) >> (64 - 2 * fc); });
					uint64_t suf = 0; Bk::download(&suf, d_small + 32, 8);
					absorb[suf] = (uint8_t)cntrun;
				}
				a = b + 1;
			}
		}
		// fchr: base counts of the text = fc-mer counts by leading base + the last fc-1 positions
		{
			uint64_t bc[4] = {0, 0, 0, 0};
			const uint64_t per = (ftab_len - 1) / 4;
			for (int c = 0; c < 4; c++) for (uint64_t k = 0; k < per; k++) bc[c] += hist[1 + (uint64_t)c * per + k];
			const uint64_t tail0 = n_ >= (uint64_t)fc ? n_ - (uint64_t)fc + 1 : 0;
			if (n_ > tail0) {
				const uint64_t nt = n_ - tail0;
				Bk::pfor(nt, [=] BT2_HD_LAMBDA (uint64_t k) { d_small[k] = (uint64_t)text_base(W, tail0 + k); });
				std::vector<uint64_t> tb(nt); Bk::download(tb.data(), d_small, nt * 8);
				for (uint64_t v : tb) bc[v]++;
			}
			out.fchr[0] = 0;
			for (int c = 0; c < 4; c++) out.fchr[c + 1] = out.fchr[c] + bc[c];
		}
		// ftab / eftab: running sum; a transition that absorbs short suffixes becomes a pointer into eftab (bt2_idx.h:3128-3150)
		{
			const uint64_t off_mask = P.off_size == 4 ? 0xffffffffull : 0xffffffffffffffffull;
			out.ftab.assign(ftab_len, 0);
			out.eftab.assign((size_t)fc * 2, 0);
			uint64_t hi_prev = 0, ecur = 0;
			for (uint64_t i = 1; i < ftab_len; i++) {
				const uint64_t lo = hist[i] + hi_prev;
				if (absorb[i] > 0) {
					const uint64_t hi = lo + absorb[i];
					if (ecur * 2 + 1 >= out.eftab.size()) { free_all(); err = "eftab overflow"; return false; }
					out.eftab[ecur * 2] = lo; out.eftab[ecur * 2 + 1] = hi;
					out.ftab[i] = (ecur++) ^ off_mask;
					hi_prev = hi;
				} else { out.ftab[i] = lo; hi_prev = lo; }
			}
		}
		free_all();
		return true;
	}
};

} } // namespace bt2g::build
#endif
