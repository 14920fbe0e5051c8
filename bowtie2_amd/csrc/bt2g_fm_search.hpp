// bt2g_fm_search.hpp -- the pure FM-index phases of the worker as free functions.
//
// "Pure" = their result depends only on the read and the index, not on RNG state, limits or
// what has been reported so far.  That makes them batchable: bt2g_kernels.hip runs them one
// lane per task for a whole batch (thousands of independent backward-search chains in flight,
// the HBM-bound shape), and the per-read worker consumes the results in the reference's order.
// The same functions are also called inline by the worker for the cases that are not
// pre-computed (re-seeding rounds, buffer overflow), so both paths share one implementation.
//
//   fm_extend_hit   SwDriver::extend            aligner_sw_driver.cpp:299-484
//   fm_one_mm_dir   SeedAligner::oneMmSearch    aligner_seed.cpp:975-1325 (one fw/rc x BWT/BWT' combination)
#ifndef BT2G_FM_SEARCH_HPP_
#define BT2G_FM_SEARCH_HPP_

#include "bt2g_device.hpp"
#include "../../include/bt2g.h"

namespace bt2g {

struct FmCount { uint32_t bwops, sides; };

BT2_HD int fm_comp(int c) { return c < 4 ? 3 - c : 4; }

// mapBiLFEx (bt2_idx.h:2372): ranks of all four characters at top and bot of `e`, and the prefix
// sums that update the range in the other index, starting at topp.
template <typename TOff>
BT2_HD void fm_bi_lf(const DevEbwt<TOff>& e, TOff top, TOff bot, TOff topp, TOff t[4], TOff b[4], TOff tp[4], TOff bp[4], FmCount& cnt) {
	cnt.sides += (uint32_t)rank4_pair(e, top, bot, t, b);
	tp[0] = topp;
	bp[0] = tp[0] + (b[0] - t[0]);
	tp[1] = bp[0]; bp[1] = tp[1] + (b[1] - t[1]);
	tp[2] = bp[1]; bp[2] = tp[2] + (b[2] - t[2]);
	tp[3] = bp[2]; bp[3] = tp[3] + (b[3] - t[3]);
}

// RD: struct with `int seq(uint32_t i) const` (fw read char, codes 0..4) and `int qual(uint32_t i) const` (ASCII)
template <typename RD>
BT2_HD int fm_rd_char(const RD& rd, uint32_t rdlen, bool fw, uint32_t i) { return fw ? rd.seq(i) : fm_comp(rd.seq(rdlen - 1 - i)); }

// Extend a seed hit left (forward index) and right (mirror index) while the SA range keeps its
// size and the read agrees; at most 255 positions each way.
template <typename TOff, typename RD>
BT2_HD void fm_extend_hit(const DevIndex<TOff>& ix, const RD& rd, uint32_t rdlen, TOff topf, TOff botf, TOff topb, TOff botb,
                          bool fw, uint32_t off, uint32_t len, uint32_t& nlex, uint32_t& nrex, FmCount& cnt, bool right = true) {
	TOff t[4], b[4], tp[4], bp[4];
	nlex = nrex = 0;
	for (int side = 0; side < 2; side++) {
		const bool left = side == 0;
		if (!left && !right) continue;     // the reference only extends to the right when the mirror index is loaded (:403)
		const uint32_t lim = left ? (fw ? off : rdlen - len - off) : (fw ? rdlen - len - off : off);
		if (lim == 0) continue;
		const DevEbwt<TOff>& e = left ? ix.fw : ix.bw;
		TOff top = left ? topf : topb, bot = left ? botf : botb;
		const TOff other = left ? topb : topf;
		uint32_t n = 0;
		for (uint32_t ii = 0; ii < lim; ii++) {
			uint32_t i;
			if (left) i = fw ? off - ii - 1 : rdlen - off - len - 1 - ii;
			else      i = fw ? ii + len + off : rdlen - off + ii;
			const int rdc = fm_rd_char(rd, rdlen, fw, i);
			if (bot - top > 1) {
				cnt.bwops++;
				fm_bi_lf(e, top, bot, other, t, b, tp, bp, cnt);
				int nonz = -1;
				bool abort = false;
				const TOff orig = bot - top;
				for (int j = 0; j < 4; j++) {
					if (b[j] > t[j]) {
						if (nonz >= 0) { abort = true; break; }
						nonz = j; top = t[j]; bot = b[j];
					}
				}
				if (abort || (nonz != rdc && rdc <= 3) || bot - top < orig) break;
			} else {
				cnt.bwops++;
				if (top != e.zoff) cnt.sides++;
				TOff row = top;
				const int c = map_lf1(e, row);
				top = row;
				if (c != rdc && rdc <= 3) break;
				bot = top + 1;
			}
			if (++n == 255) break;
		}
		if (left) nlex = n; else nrex = n;
	}
}

// The joined text (every unambiguous reference base, in index order) is what <base>.4 stores, two bits per base.
BT2_HD int joined_char(const DevRef& r, uint64_t p) { return (r.buf[p >> 2] >> ((p & 3) << 1)) & 3; }

// ---------------------------------------------------------------------------------------------------------------------
// SwDriver::extend for a range of 1..kExtRows rows whose joined-text offsets are known, sixteen characters per step.
//
// While the walk of fm_extend_hit keeps a range of several rows, every row of it is preceded (left, forward index) or followed
// (right, mirror index) by the same character -- that is what "the range keeps its size" means -- and that character equals the
// read's unless the read has an N there.  In text terms: position ii is accepted iff the rows' texts all hold the same character
// there and (read N or character == read character); a row that has run out of text (the '$' row leaves the range) ends it.  For
// one row: out of text (mapLF1 fails at either end of the joined text) counts as a mismatch, which a read N forgives, exactly as in the walk.
// The LF walk is a chain of dependent rank queries, up to 255 long on either side, and one such chain in a wave of short ones keeps
// the whole workgroup resident for its duration (identical segmental-duplication copies: 2-row ranges that extend for a hundred
// characters); here the only dependent loads are the rows' suffix-array entries.
// Sixteen characters are packed two bits each into one word per row and one for the read (RD::window16 -> multiply trick), XORed, and
// the first set bit pair is the first position that ends the extension.
constexpr uint32_t kExtRows = 8;
// the four 2-bit codes in the bytes of x (values 0..3; bit 2 = N) -> bits 0-7; and their N flags -> bits 0-3
BT2_HD uint32_t pack4_codes(uint32_t x) { return (uint32_t)(((x & 0x03030303u) * 0x01041040u) >> 24); }
BT2_HD uint32_t pack4_nflags(uint32_t x) { return (uint32_t)((((x >> 2) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu; }
// reverse the order of the sixteen 2-bit groups of a word
BT2_HD uint32_t rev_groups(uint32_t x) {
	x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
	x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
	x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
	return (x >> 16) | (x << 16);
}
BT2_HD uint32_t rev16(uint32_t x) {
	x = ((x >> 1) & 0x5555u) | ((x & 0x5555u) << 1); x = ((x >> 2) & 0x3333u) | ((x & 0x3333u) << 2);
	x = ((x >> 4) & 0x0f0fu) | ((x & 0x0f0fu) << 4); return ((x >> 8) | (x << 8)) & 0xffffu;
}
// bit i of a 16-bit mask -> both bits of group i
BT2_HD uint32_t spread_groups(uint32_t m) {
	m = (m | (m << 8)) & 0x00ff00ffu; m = (m | (m << 4)) & 0x0f0f0f0fu; m = (m | (m << 2)) & 0x33333333u; m = (m | (m << 1)) & 0x55555555u;
	return m * 3u;
}
// sixteen joined-text characters starting at position q (ascending), two bits each, first character in bits 0-1; positions outside
// [0, n) read as 0 and are flagged in `bad` (one bit per position)
BT2_HD uint32_t text16(const DevRef& r, uint64_t n, int64_t q, uint32_t& bad) {
	bad = 0;
	if (q < 0) { const uint32_t k = (uint32_t)(-q); if (k >= 16u) { bad = 0xffffu; return 0; } bad = (1u << k) - 1u; }
	if (q + 16 > (int64_t)n) { const int64_t ok = (int64_t)n - q; if (ok <= 0) { bad = 0xffffu; return 0; } bad |= (0xffffu << (uint32_t)ok) & 0xffffu; }
	const uint64_t q0 = q < 0 ? 0 : (uint64_t)q;           // first position that exists
	const uint64_t nbytes = (n + 3) >> 2;
	uint64_t b0 = q0 >> 2;
	uint64_t v = 0;
	if (b0 + 8 <= nbytes) __builtin_memcpy(&v, r.buf + b0, 8);
	else for (uint32_t k = 0; b0 + k < nbytes && k < 8; k++) v |= (uint64_t)r.buf[b0 + k] << (8 * k);
	uint32_t w = (uint32_t)(v >> ((q0 & 3) << 1));        // characters q0, q0+1, ... (40 bits held: 16 characters + 3 of slack)
	if (q < 0) w <<= 2u * (uint32_t)(-q);                 // (fewer than 16 missing)
	return w & ~spread_groups(bad);
}
template <typename TOff, typename RD>
BT2_HD void fm_extend_rows_text(const DevIndex<TOff>& ix, const RD& rd, uint32_t rdlen, const uint64_t (&p)[kExtRows], uint32_t nrows, bool fw, uint32_t off, uint32_t len,
                                uint32_t& nlex, uint32_t& nrex, uint32_t& steps_walked, bool right = true) {
	const uint64_t n = (uint64_t)ix.fw.len;
	nlex = nrex = 0; steps_walked = 0;
	for (int side = 0; side < 2; side++) {
		const bool left = side == 0;
		if (!left && !right) continue;
		const uint32_t lim = left ? (fw ? off : rdlen - len - off) : (fw ? rdlen - len - off : off);
		// In the coordinates of the read as stored (5' -> 3'), the walk goes DOWN from off - 1 (left of a fw seed, right of an rc seed) or
		// UP from off + len (right of a fw seed, left of an rc seed); an rc seed compares complemented characters.
		const bool down = left == fw;
		uint32_t cnt = 0;
		bool stop = false;
		for (uint32_t i0 = 0; i0 < lim && !stop; i0 += 16) {
			// ---- the read's sixteen characters, position ii = i0 + k in bit pair k ----
			uint32_t w4[4];
			const uint32_t lo = down ? (off - i0 >= 16u ? off - i0 - 16u : 0u) : off + len + i0;      // first stored position of the window
			rd.window16(lo, w4);
			uint32_t R = pack4_codes(w4[0]) | (pack4_codes(w4[1]) << 8) | (pack4_codes(w4[2]) << 16) | (pack4_codes(w4[3]) << 24);
			uint32_t N = pack4_nflags(w4[0]) | (pack4_nflags(w4[1]) << 4) | (pack4_nflags(w4[2]) << 8) | (pack4_nflags(w4[3]) << 12);
			if (down) {
				// the window ends at stored position off - i0 - 1 = ii i0; when fewer than 16 characters are left it starts at 0 and its
				// top groups lie beyond the seed's side: shift them out so that ii i0 sits in the top group before reversing
				const uint32_t have = off - i0 >= 16u ? 16u : off - i0;
				R <<= 2u * (16u - have); N = (N << (16u - have)) & 0xffffu;
				R = rev_groups(R);
				N = rev16(N);
			}
			if (!fw) R = ~R;
			const uint32_t N2 = spread_groups(N);
			// ---- the rows' texts ----
			uint32_t mism = 0, T0 = 0;
#pragma unroll
			for (uint32_t r = 0; r < kExtRows; r++) {
				if (r >= nrows) continue;
				uint32_t bad;
				uint32_t T;
				if (left) { T = text16(ix.ref, n, (int64_t)p[r] - (int64_t)i0 - 16, bad); T = rev_groups(T); bad = rev16(bad); }
				else T = text16(ix.ref, n, (int64_t)(p[r] + len + i0), bad);
				const uint32_t bad2 = spread_groups(bad);
				if (nrows == 1) mism |= (((T ^ R) | bad2) & ~N2);        // out of text: a mismatch that a read N forgives (mapLF1 at the '$' row)
				else {
					mism |= ((T ^ R) & ~N2) | bad2;                         // a row out of text leaves the range
					if (r == 0) T0 = T; else mism |= T ^ T0;                // the rows must agree with each other, read N or not
				}
			}
			// groups: any set bit in a pair ends the extension there
			const uint32_t g = (mism | (mism >> 1)) & 0x55555555u;
			uint32_t k = g ? (uint32_t)__builtin_ctz(g) >> 1 : 16u;
			if (i0 + k > lim) k = lim - i0;
			if (k < 16u && i0 + k < lim) stop = true;
			if (cnt + k >= 255u) { cnt = 255u; stop = true; break; }
			cnt += k;
		}
		steps_walked += cnt + ((stop && cnt < 255u) ? 1u : 0u);
		if (left) nlex = cnt; else nrex = cnt;
	}
}

// (cached offset resolution of a one-row seed hit: joff_pack / kJoffNone, bt2g_device.hpp)

// 1-mismatch end-to-end hit as oneMmSearch reports it (EEHit with one Edit)
struct Mm1Hit {
	uint64_t top, bot;
	int32_t  score;
	uint16_t epos;         // offset from the 5' end
	uint8_t  echr, eqchr;  // reference char / read char, codes 0..4
};

// Batch-wide results of the pure FM phases, computed by lane-per-task kernels before the fused
// worker runs (bt2g_kernels.hip).  Any pointer may be null: the worker then computes that phase itself.
struct PreComp {
	const BT2_G bt2g_sweep_out* sweep;   // [n_reads]                       exactSweep
	const BT2_G bt2g_seed_hit*  seeds;   // [n_reads][2][max_seeds]         seed round 0 (offset 0)
	const BT2_G uint32_t*       ext;     // [n_reads][2][max_seeds]         nlex | nrex << 16 of each non-empty seed hit
	const BT2_G uint64_t*       joff;    // [n_reads][2][max_seeds]         joff_pack() of every one-row seed hit (kJoffNone otherwise): resolved while extending
	const BT2_G Mm1Hit*         mm1;     // [n_reads][2 strands][2 dirs][mm1_cap]
	const BT2_G uint8_t*        mm1_n;   // [n_reads][4]   hits per list; 255 = list overflowed
	uint32_t max_seeds, mm1_cap;
	// re-seeding rounds 1..kMaxPreRounds-1 (bt2_search.cpp:3881-4160: same seeds shifted by interval*round/nrounds), computed
	// only for reads whose previous round averaged >= seed_boost_thresh hits per seed -- the one condition for a further round
	// that does not depend on what the worker has reported by then.  [round][...] with the layout of seeds/ext/joff above.
	const BT2_G bt2g_seed_hit*  seeds_r[4];
	const BT2_G uint32_t*       ext_r[4];
	const BT2_G uint64_t*       joff_r[4];
};
constexpr uint32_t kMaxPreRounds = 4;

// offset of re-seeding round `roundi` for a read (multiseedSearchWorker, bt2_search.cpp:3885-3930); false = the round does not run
BT2_HD bool reseed_offset(uint32_t roundi, uint32_t n_seed_rounds, uint32_t interval, uint32_t seedlen, uint32_t len, uint32_t& offset) {
	uint32_t nrounds = n_seed_rounds;
	if (nrounds > interval) nrounds = interval;
	if (roundi >= nrounds || interval <= roundi) return false;
	offset = (interval * roundi) / nrounds;
	if (offset > 0 && seedlen + offset > len) return false;
	return true;
}

// The continuation of one 1-mismatch branch of oneMmSearch (aligner_seed.cpp:1189-1300): position `dep` (from the 3' end of the
// search direction) was matched with reference character j instead of the read's; the rest of the read must match exactly.
// emit() is called if it does and the hit is valid.  A free function so that the batch kernels can run the branches of
// all reads as one flat list of tasks (bt2g_kernels.hip: k_one_mm_cont) instead of nested, divergent loops.
template <typename TOff, typename RD, typename Emit>
BT2_HD void fm_one_mm_cont(const DevIndex<TOff>& ix, const bt2g_align_params& P, int64_t minsc, const RD& rd, uint32_t len, bool fw, bool ebwtfw,
                           uint32_t dep, int j, TOff topm, TOff botm, TOff topmp, TOff botmp, Emit emit, FmCount& cnt) {
	constexpr TOff kMask = (TOff)OffTraits<TOff>::kMask;
	const DevEbwt<TOff>& e = ebwtfw ? ix.fw : ix.bw;
	auto sq = [&](uint32_t i) -> int {
		if (fw) return ebwtfw ? rd.seq(i) : rd.seq(len - 1 - i);
		return ebwtfw ? fm_comp(rd.seq(len - 1 - i)) : fm_comp(rd.seq(i));
	};
	auto ql = [&](uint32_t i) -> int {
		const bool rev = fw ? !ebwtfw : ebwtfw;
		return rev ? rd.qual(len - 1 - i) : rd.qual(i);
	};
	const int rdc = sq(len - dep - 1);
	const int quc = ql(len - dep - 1);
	uint32_t depm = dep + 1;
	TOff tm[4], bm[4], tmp[4], bmp[4];
	for (; depm < len; depm++) {
		const int rdcm = sq(len - depm - 1);
		if (botm - topm > 1) {
			cnt.bwops++;
			fm_bi_lf(e, topm, botm, topmp, tm, bm, tmp, bmp, cnt);
			if (rdcm > 3) { topm = botm = 0; break; }
			topm = tm[rdcm]; botm = bm[rdcm];
			topmp = tmp[rdcm]; botmp = bmp[rdcm];
			if (botm <= topm) break;
		} else {
			cnt.bwops++; cnt.sides++;
			topm = map_lf1c(e, topm, rdcm);
			if (topm == kMask) break;
			botm = topm + 1;
		}
	}
	if (depm != len) return;
	uint32_t off5p = dep;
	if (fw == ebwtfw) off5p = len - off5p - 1;
	int64_t score = (int64_t)(len - 1) * P.match_bonus;
	int pen;   // Scoring::score(rdc, 1<<j, quc-33)
	{
		int q = quc - 33; if (q < 0) q = 0; if (q > 255) q = 255;
		if (rdc > 3) pen = -P.n_pen;
		else if (rdc == j) pen = P.match_bonus;
		else {
			if (P.mm_type == 3) { const int qq = q < 40 ? q : 40; const float frac = (float)qq / 40.0f; pen = -(P.mm_min + (int)(frac * (float)(P.mm_max - P.mm_min))); }
			else if (P.mm_type == 2) pen = -(q < 5 ? 0 : (q < 15 ? 10 : (q < 25 ? 20 : 30)));
			else pen = -P.mm_max;
		}
	}
	score += pen;
	bool valid = true;
	if (P.match_bonus > 0) {
		// --local: the end-to-end hit must also be a legal local alignment, i.e. its running score may not
		// touch 0 at the mismatch from either end (aligner_seed.cpp:1231-1260)
		int64_t fwsc = 0, bwsc = 0;
		for (uint32_t i = 0; i < len; i++) {
			if (i == dep) { if (fwsc + pen <= 0) { valid = false; break; } fwsc += pen; } else fwsc += P.match_bonus;
			if (len - i - 1 == dep) { if (bwsc + pen <= 0) { valid = false; break; } bwsc += pen; } else bwsc += P.match_bonus;
		}
	}
	if (valid && score >= minsc) {
		Mm1Hit h;
		h.top = ebwtfw ? (uint64_t)topm : (uint64_t)topmp;
		h.bot = ebwtfw ? (uint64_t)botm : (uint64_t)botmp;
		h.score = (int32_t)score; h.epos = (uint16_t)off5p; h.echr = (uint8_t)j; h.eqchr = (uint8_t)rdc;
		emit(h);
	}
}

// One (read strand, index direction) combination of oneMmSearch with repex=false, rep1mm=true.
// emit(const Mm1Hit&) is called for every valid 1-mismatch end-to-end hit, in discovery order.
// defer(dep, j, top, bot, topp, botp) -> bool may take a branch's continuation away (the batch kernels queue it); when it
// returns false the branch is finished here.
struct Mm1NoDefer { template <typename TOff> BT2_HD bool operator()(uint32_t, int, TOff, TOff, TOff, TOff) const { return false; } };
template <typename TOff, typename RD, typename Emit, typename Defer = Mm1NoDefer>
BT2_HD void fm_one_mm_dir(const DevIndex<TOff>& ix, const bt2g_align_params& P, int64_t minsc, int nceil,
                          const RD& rd, uint32_t len, uint32_t ns, bool fw, bool ebwtfw, Emit emit, FmCount& cnt, Defer defer = Defer()) {
	constexpr TOff kMask = (TOff)OffTraits<TOff>::kMask;
	const DevEbwt<TOff>& e = ebwtfw ? ix.fw : ix.bw;
	const DevEbwt<TOff>& ep = ebwtfw ? ix.bw : ix.fw;
	// seq views (aligner_seed.cpp:1031-1040): fw: patFw | patFwRev ; rc: patRc | patRcRev
	auto sq = [&](uint32_t i) -> int {
		if (fw) return ebwtfw ? rd.seq(i) : rd.seq(len - 1 - i);
		return ebwtfw ? fm_comp(rd.seq(len - 1 - i)) : fm_comp(rd.seq(i));
	};
	const uint32_t halfFw = len >> 1;
	const uint32_t halfBw = (len >> 1) + ((len & 1) ? 1 : 0);
	const uint32_t ftab_len = e.ftab_chars;
	const uint32_t nea = ebwtfw ? halfFw : halfBw;
	for (uint32_t dep = 0; dep < nea; dep++) if (sq(len - dep - 1) > 3) return;
	TOff t[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, tp[4] = {0, 0, 0, 0}, bp[4] = {0, 0, 0, 0};
	uint32_t dep = 0;
	TOff top = 0, bot = 0, topp = 0, botp = 0;
	if (ftab_len > 1 && ftab_len <= nea) {
		uint64_t kt = 0, kr = 0;   // the window seq[len-ftab, len): text order in `e`, reversed in `ep`
		for (uint32_t i = 0; i < ftab_len; i++) {
			kt = (kt << 2) | (uint64_t)sq(len - ftab_len + i);
			kr = (kr << 2) | (uint64_t)sq(len - 1 - i);
		}
		top = ftab_hi(e, kt); bot = ftab_lo(e, kt + 1);
		topp = ftab_hi(ep, kr); botp = ftab_lo(ep, kr + 1);
		if (bot - top == 0) return;
		dep = ftab_len;
	} else {
		const int c = sq(len - 1);
		top = topp = e.fchr[c];
		bot = botp = e.fchr[c + 1];
		if (bot - top == 0) return;
		dep = 1;
	}
	for (; dep < nea; dep++) {
		const int rdc = sq(len - dep - 1);
		if (bot - top > 1) {
			cnt.bwops++;
			fm_bi_lf(e, top, bot, topp, t, b, tp, bp, cnt);
			top = t[rdc]; bot = b[rdc];
			if (bot <= top) return;
			topp = tp[rdc]; botp = bp[rdc];
		} else {
			cnt.bwops++; cnt.sides++;
			top = map_lf1c(e, top, rdc);
			if (top == kMask) return;
			bot = top + 1;
		}
	}
	for (; dep < len; dep++) {
		const int rdc = sq(len - dep - 1);
		if (rdc > 3 && nceil == 0) break;
		int clo = 0, chi = 3;
		bool match = true;
		if (bot - top > 1) {
			cnt.bwops++;
			fm_bi_lf(e, top, bot, topp, t, b, tp, bp, cnt);
			match = rdc < 4;
			if (match) { top = t[rdc]; bot = b[rdc]; topp = tp[rdc]; botp = bp[rdc]; }
			else { top = bot = 0; }
		} else {
			cnt.bwops++;
			if (top != e.zoff) cnt.sides++;
			TOff row = top;
			clo = map_lf1(e, row);
			match = (clo == rdc);
			if (clo < 0) break;
			top = row;
			t[clo] = top; b[clo] = bot = top + 1;
			bp[clo] = botp; tp[clo] = topp;
			chi = clo;
		}
		if (ns == 0 || rdc > 3) {
			for (int j = clo; j <= chi; j++) {
				if (j == rdc || b[j] == t[j]) continue;
				if (dep + 1 < len && defer(dep, j, t[j], b[j], tp[j], bp[j])) continue;
				fm_one_mm_cont(ix, P, minsc, rd, len, fw, ebwtfw, dep, j, t[j], b[j], tp[j], bp[j], emit, cnt);
			}
		}
		if (bot > top && match) {
			if (dep == len - 1) break;   // exact end-to-end hit: not reported here (repex = false)
		} else {
			break;
		}
	}
}

} // namespace bt2g
#endif
