// bt2g_device.hpp -- device-side view of the FM index and the rank/LF primitives.
//
// HBM layout: the .1.bt2[l] "sides" are kept exactly as on disk -- each side is one
// 64-byte (.bt2) or 128-byte (.bt2l) line holding 48/96 bytes of 2-bit BWT followed by
// occ[A,C,G,T] -- so one rank query is one aligned line read (SURVEY.md section 8d:
// algorithmic bytes per rank query = side_sz).  ftab/eftab/offs/rstarts/plen are uploaded
// verbatim at index width TOff (uint32_t for .bt2, uint64_t for .bt2l).
//
// Reference semantics restated here (never copied):
//   rank  = Ebwt::countBt2Side / countBt2SideEx   bt2_idx.h:1758,1887
//   LF    = Ebwt::mapLF / mapLF1 / mapBiLFEx       bt2_idx.h:2313-2473
//   ftab  = Ebwt::ftabLoHi / ftabSeqToInt          bt2_idx.h:1374-1554
//   SA    = Ebwt::getOffset / joinedToTextOff      bt2_idx.cpp:54-171
#ifndef BT2G_DEVICE_HPP_
#define BT2G_DEVICE_HPP_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BT2_HD __host__ __device__ __forceinline__
#ifdef BT2G_INLINE_ALL
#define BT2_HDN __host__ __device__ __forceinline__
#else
#define BT2_HDN __host__ __device__ __attribute__((noinline))   // large phase functions: real calls keep register pressure local
#endif
#define BT2_D __device__ __forceinline__
#else
#define BT2_HD inline
#define BT2_HDN
#define BT2_D inline
#endif

#if !defined(__HIPCC__)
struct ulonglong2 { unsigned long long x, y; };   // host-side stand-in (test builds only)
#endif

// Scalar loads / stores for pointers that are KNOWN to point into the wave's arena in HBM.  Through a plain (generic)
// pointer the compiler must issue FLAT instructions: it then has to assume the access may alias LDS (so every LDS value it
// holds in registers is reloaded after a store) and the wait for a FLAT load also drains the LDS counter.  The device
// version names the global address space; the host twin (tests/hostsim) is an ordinary dereference.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> __host__ __device__ __forceinline__ T gld(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }
template <typename T> __host__ __device__ __forceinline__ void gst(T* p, T v) { *(__attribute__((address_space(1))) T*)p = v; }
#else
template <typename T> BT2_HD T gld(const T* p) { return *p; }
template <typename T> BT2_HD void gst(T* p, T v) { *p = v; }
#endif

namespace bt2g {

template <typename TOff> struct OffTraits;
template <> struct OffTraits<uint32_t> {
	static constexpr uint32_t kSideSz = 64, kSideBwtLen = 192, kBwtWords = 6; // 48 B of BWT = 6 x u64
	static constexpr uint32_t kMask = 0xffffffffu;
};
template <> struct OffTraits<uint64_t> {
	static constexpr uint32_t kSideSz = 128, kSideBwtLen = 384, kBwtWords = 12; // 96 B of BWT = 12 x u64
	static constexpr uint64_t kMask = ~0ull;
};

template <typename TOff>
struct DevEbwt {
	const uint8_t* ebwt;   // sides
	const TOff*    ftab;
	const TOff*    eftab;
	const TOff*    offs;   // SA sample (forward index only)
	TOff len, zoff;
	TOff fchr[5];
	uint32_t ftab_chars, off_rate;
	uint32_t is_fw;
};

struct DevRef {
	const uint64_t* rec_refpos;   // [nrecs] start of stretch within its reference (incl. Ns)
	const uint64_t* rec_bufpos;   // [nrecs] start of stretch within the 2-bit buffer
	const uint64_t* rec_len;      // [nrecs]
	const uint64_t* ref_rec_offs; // [nrefs+1]
	const uint64_t* ref_lens;     // [nrefs]
	const uint8_t*  buf;          // 2-bit packed, 4 bases/byte, LSB first
	uint64_t nrefs;
};

template <typename TOff>
struct DevIndex {
	DevEbwt<TOff> fw, bw;
	const TOff* rstarts;  // [3*n_frag]
	const TOff* plen;     // [n_pat]
	TOff n_frag, n_pat;
	DevRef ref;
};

struct DevCounters {
	unsigned long long rank_queries, sa_lookups, ftab_lookups, dp_cells, bwops;
};

// ---------------------------------------------------------------------------------------
// 2-bit counting.  A u64 word holds 32 BWT characters, char i at bits [2i,2i+1].
// eq_mask(w,c): bit 2i set iff char i == c.
BT2_HD uint64_t eq_mask(uint64_t w, int c) {
	// XOR with the complement pattern of c so that matching pairs become 0b11
	const uint64_t pat = (c == 0) ? ~0ull : (c == 1) ? 0xaaaaaaaaaaaaaaaaull : (c == 2) ? 0x5555555555555555ull : 0ull;
	const uint64_t x = w ^ pat;
	return x & (x >> 1) & 0x5555555555555555ull;
}

BT2_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __popcll(x);
#else
	return __builtin_popcountll(x);
#endif
}

// Low-bit mask selecting the first n (0..32) characters of a word, on the 0x55.. lattice.
BT2_HD uint64_t first_chars_mask(uint32_t n) {
	return n >= 32 ? 0x5555555555555555ull : (((1ull << (2 * n)) - 1ull) & 0x5555555555555555ull);
}

// One side, loaded into registers.  kBwtWords u64 of BWT + 4 occ counters.
template <typename TOff>
struct Side {
	uint64_t w[OffTraits<TOff>::kBwtWords];
	TOff occ[4];
};

template <typename TOff>
BT2_HD void load_side(const uint8_t* ebwt, uint64_t side_num, Side<TOff>& s) {
	constexpr uint32_t NW = OffTraits<TOff>::kBwtWords;
	// 16-byte vector loads: 4 per .bt2 side, 8 per .bt2l side -- one full aligned line
	const ulonglong2* p = reinterpret_cast<const ulonglong2*>(ebwt + side_num * OffTraits<TOff>::kSideSz);
#pragma unroll
	for (uint32_t i = 0; i < NW / 2; i++) {
		ulonglong2 v = p[i];
		s.w[2 * i] = v.x;
		s.w[2 * i + 1] = v.y;
	}
	if (sizeof(TOff) == 4) {
		ulonglong2 v = p[NW / 2];
		s.occ[0] = (TOff)(v.x & 0xffffffffull); s.occ[1] = (TOff)(v.x >> 32);
		s.occ[2] = (TOff)(v.y & 0xffffffffull); s.occ[3] = (TOff)(v.y >> 32);
	} else {
		ulonglong2 v0 = p[NW / 2], v1 = p[NW / 2 + 1];
		s.occ[0] = (TOff)v0.x; s.occ[1] = (TOff)v0.y; s.occ[2] = (TOff)v1.x; s.occ[3] = (TOff)v1.y;
	}
}

// # of chars == c among the first char_off chars of the side
template <typename TOff>
BT2_HD uint32_t side_count(const Side<TOff>& s, uint32_t char_off, int c) {
	constexpr uint32_t NW = OffTraits<TOff>::kBwtWords;
	uint32_t cnt = 0;
#pragma unroll
	for (uint32_t i = 0; i < NW; i++) {
		const uint32_t lo = i * 32;
		const uint32_t n = char_off > lo ? char_off - lo : 0;  // chars of this word that count
		cnt += popc64(eq_mask(s.w[i], c) & first_chars_mask(n));
	}
	return cnt;
}

template <typename TOff>
BT2_HD int side_char(const Side<TOff>& s, uint32_t char_off) {
	constexpr uint32_t NW = OffTraits<TOff>::kBwtWords;
	uint64_t w = 0;
#pragma unroll
	for (uint32_t i = 0; i < NW; i++) if ((char_off >> 5) == i) w = s.w[i];
	return (int)((w >> ((char_off & 31) * 2)) & 3);
}

// rank_c(row) = fchr[c] + occ_side[c] + count - ['$' fix]   (countBt2Side, bt2_idx.h:1758)
template <typename TOff>
BT2_HD TOff rank_in_side(const DevEbwt<TOff>& e, const Side<TOff>& s, uint64_t side_num, uint32_t char_off, int c) {
	uint32_t cnt = side_count(s, char_off, c);
	if (c == 0) {
		const uint64_t zside = (uint64_t)e.zoff / OffTraits<TOff>::kSideBwtLen;
		const uint32_t zchar = (uint32_t)((uint64_t)e.zoff % OffTraits<TOff>::kSideBwtLen);
		if (side_num == zside && char_off > zchar) cnt--;   // '$' is stored as 'A' (bt2_idx.h:1766-1774)
	}
	return (TOff)(e.fchr[c] + s.occ[c] + cnt);
}

template <typename TOff>
BT2_HD TOff rank1(const DevEbwt<TOff>& e, TOff row, int c) {
	const uint64_t side_num = (uint64_t)row / OffTraits<TOff>::kSideBwtLen;
	const uint32_t char_off = (uint32_t)((uint64_t)row % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, side_num, s);
	return rank_in_side(e, s, side_num, char_off, c);
}

// all four characters at once (countBt2SideEx, bt2_idx.h:1887)
template <typename TOff>
BT2_HD void rank4_in_side(const DevEbwt<TOff>& e, const Side<TOff>& s, uint64_t side_num, uint32_t char_off, TOff out[4]) {
	uint32_t c1 = side_count(s, char_off, 1), c2 = side_count(s, char_off, 2), c3 = side_count(s, char_off, 3);
	uint32_t c0 = char_off - c1 - c2 - c3;
	const uint64_t zside = (uint64_t)e.zoff / OffTraits<TOff>::kSideBwtLen;
	const uint32_t zchar = (uint32_t)((uint64_t)e.zoff % OffTraits<TOff>::kSideBwtLen);
	if (side_num == zside && char_off > zchar) c0--;
	out[0] = (TOff)(e.fchr[0] + s.occ[0] + c0);
	out[1] = (TOff)(e.fchr[1] + s.occ[1] + c1);
	out[2] = (TOff)(e.fchr[2] + s.occ[2] + c2);
	out[3] = (TOff)(e.fchr[3] + s.occ[3] + c3);
}

template <typename TOff>
BT2_HD void rank4(const DevEbwt<TOff>& e, TOff row, TOff out[4]) {
	const uint64_t side_num = (uint64_t)row / OffTraits<TOff>::kSideBwtLen;
	const uint32_t char_off = (uint32_t)((uint64_t)row % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, side_num, s);
	rank4_in_side(e, s, side_num, char_off, out);
}

// Pair query rank_c(top), rank_c(bot): one side read when both loci share a side
// (SideLocus::initFromTopBot, bt2_idx.h:325-348), else two.  Returns # sides read.
template <typename TOff>
BT2_HD int rank1_pair(const DevEbwt<TOff>& e, TOff top, TOff bot, int c, TOff& otop, TOff& obot) {
	const uint64_t st = (uint64_t)top / OffTraits<TOff>::kSideBwtLen, sb = (uint64_t)bot / OffTraits<TOff>::kSideBwtLen;
	const uint32_t ct = (uint32_t)((uint64_t)top % OffTraits<TOff>::kSideBwtLen), cb = (uint32_t)((uint64_t)bot % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, st, s);
	otop = rank_in_side(e, s, st, ct, c);
	if (sb == st) { obot = rank_in_side(e, s, sb, cb, c); return 1; }
	load_side<TOff>(e.ebwt, sb, s);
	obot = rank_in_side(e, s, sb, cb, c);
	return 2;
}

template <typename TOff>
BT2_HD int rank4_pair(const DevEbwt<TOff>& e, TOff top, TOff bot, TOff t[4], TOff b[4]) {
	const uint64_t st = (uint64_t)top / OffTraits<TOff>::kSideBwtLen, sb = (uint64_t)bot / OffTraits<TOff>::kSideBwtLen;
	const uint32_t ct = (uint32_t)((uint64_t)top % OffTraits<TOff>::kSideBwtLen), cb = (uint32_t)((uint64_t)bot % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, st, s);
	rank4_in_side(e, s, st, ct, t);
	if (sb == st) { rank4_in_side(e, s, sb, cb, b); return 1; }
	load_side<TOff>(e.ebwt, sb, s);
	rank4_in_side(e, s, sb, cb, b);
	return 2;
}

// mapLF1(row, l, c) (bt2_idx.h:2420): all-ones if BWT[row] != c or row is the '$' row
template <typename TOff>
BT2_HD TOff map_lf1c(const DevEbwt<TOff>& e, TOff row, int c) {
	const uint64_t side_num = (uint64_t)row / OffTraits<TOff>::kSideBwtLen;
	const uint32_t char_off = (uint32_t)((uint64_t)row % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, side_num, s);
	if (side_char(s, char_off) != c || row == e.zoff) return (TOff)OffTraits<TOff>::kMask;
	return rank_in_side(e, s, side_num, char_off, c);
}

// mapLF1(row&, l) (bt2_idx.h:2451): returns BWT char (or -1 at '$') and advances row
template <typename TOff>
BT2_HD int map_lf1(const DevEbwt<TOff>& e, TOff& row) {
	if (row == e.zoff) return -1;
	const uint64_t side_num = (uint64_t)row / OffTraits<TOff>::kSideBwtLen;
	const uint32_t char_off = (uint32_t)((uint64_t)row % OffTraits<TOff>::kSideBwtLen);
	Side<TOff> s;
	load_side<TOff>(e.ebwt, side_num, s);
	const int c = side_char(s, char_off);
	row = rank_in_side(e, s, side_num, char_off, c);
	return c;
}

// ftabHi / ftabLo with the eftab indirection for entries > len (bt2_idx.h:1428-1554)
template <typename TOff>
BT2_HD TOff ftab_hi(const DevEbwt<TOff>& e, uint64_t i) {
	const TOff v = e.ftab[i];
	if (v <= e.len) return v;
	const TOff ef = v ^ (TOff)OffTraits<TOff>::kMask;
	return e.eftab[(uint64_t)ef * 2 + 1];
}
template <typename TOff>
BT2_HD TOff ftab_lo(const DevEbwt<TOff>& e, uint64_t i) {
	const TOff v = e.ftab[i];
	if (v <= e.len) return v;
	const TOff ef = v ^ (TOff)OffTraits<TOff>::kMask;
	return e.eftab[(uint64_t)ef * 2];
}

// Ebwt::getOffset (bt2_idx.cpp:150): LF-walk to a sampled row; nsteps = # LF steps taken
template <typename TOff>
BT2_HD TOff get_offset(const DevEbwt<TOff>& e, TOff row, uint32_t& nsteps) {
	const TOff samp_mask = (TOff)(((TOff)OffTraits<TOff>::kMask) << e.off_rate);
	TOff jumps = 0;
	for (;;) {
		if (row == e.zoff) { nsteps = (uint32_t)jumps; return jumps; }
		if ((row & samp_mask) == row) { nsteps = (uint32_t)jumps; return (TOff)(jumps + e.offs[row >> e.off_rate]); }
		// mapLF(l): rank of the row's own BWT char
		const uint64_t side_num = (uint64_t)row / OffTraits<TOff>::kSideBwtLen;
		const uint32_t char_off = (uint32_t)((uint64_t)row % OffTraits<TOff>::kSideBwtLen);
		Side<TOff> s;
		load_side<TOff>(e.ebwt, side_num, s);
		row = rank_in_side(e, s, side_num, char_off, side_char(s, char_off));
		jumps++;
	}
}

// Ebwt::joinedToTextOff (bt2_idx.cpp:54) for the forward index.  tidx = all-ones if rejected.
template <typename TOff>
BT2_HD void joined_to_text_off(const DevIndex<TOff>& ix, TOff qlen, TOff off, TOff& tidx, TOff& textoff, TOff& tlen,
                               bool reject_straddle, bool& straddled) {
	TOff top = 0, bot = ix.n_frag;
	straddled = false;
	tidx = (TOff)OffTraits<TOff>::kMask; textoff = 0; tlen = 0;
	// The reference is only ever asked about offsets inside the joined text; the row of the empty
	// suffix resolves to off == len, which no fragment contains -- report it as rejected.
	if (off >= ix.fw.len || ix.n_frag == 0) return;
	for (int iter = 0; iter < 70; iter++) {
		const TOff elt = top + ((bot - top) >> 1);
		const TOff lower = ix.rstarts[(uint64_t)elt * 3];
		const TOff upper = (elt == ix.n_frag - 1) ? ix.fw.len : ix.rstarts[((uint64_t)elt + 1) * 3];
		if (lower <= off) {
			if (upper > off) {
				if (off + qlen > upper) {
					straddled = true;
					if (reject_straddle) { tidx = (TOff)OffTraits<TOff>::kMask; textoff = 0; tlen = 0; return; }
				}
				tidx = ix.rstarts[(uint64_t)elt * 3 + 1];
				textoff = (off - lower) + ix.rstarts[(uint64_t)elt * 3 + 2];
				break;
			}
			top = elt;
		} else {
			bot = elt;
		}
	}
	if (tidx != (TOff)OffTraits<TOff>::kMask) tlen = ix.plen[tidx];
}

// BitPairReference::getBase / getStretch semantics (reference.cpp:330-579) with a binary
// search over the N-free stretch records instead of the reference's linear scan.
BT2_HD int ref_base(const DevRef& r, uint64_t tidx, int64_t toff) {
	if (toff < 0 || (uint64_t)toff >= r.ref_lens[tidx]) return 4;
	uint64_t lo = r.ref_rec_offs[tidx], hi = r.ref_rec_offs[tidx + 1];
	// last record with rec_refpos <= toff
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (r.rec_refpos[mid] <= (uint64_t)toff) lo = mid; else hi = mid;
	}
	if (hi == lo) return 4;
	const uint64_t p = r.rec_refpos[lo];
	if ((uint64_t)toff < p || (uint64_t)toff >= p + r.rec_len[lo]) return 4;
	const uint64_t bo = r.rec_bufpos[lo] + ((uint64_t)toff - p);
	return (r.buf[bo >> 2] >> ((bo & 3) << 1)) & 3;
}

// The same lookup for a run of consecutive positions: ref_rec_find() locates the record for the first position once (one
// binary search per DP window instead of one per base); ref_base_at() then only steps forward from that record.
BT2_HD uint64_t ref_rec_find(const DevRef& r, uint64_t tidx, int64_t toff) {
	uint64_t lo = r.ref_rec_offs[tidx], hi = r.ref_rec_offs[tidx + 1];
	if (toff < 0) return lo;
	while (hi - lo > 1) {
		const uint64_t mid = (lo + hi) >> 1;
		if (r.rec_refpos[mid] <= (uint64_t)toff) lo = mid; else hi = mid;
	}
	return lo;
}
BT2_HD int ref_base_at(const DevRef& r, uint64_t tidx, int64_t toff, uint64_t rec) {
	if (toff < 0 || (uint64_t)toff >= r.ref_lens[tidx]) return 4;
	const uint64_t hi = r.ref_rec_offs[tidx + 1];
	if (rec >= hi) return 4;
	while (rec + 1 < hi && r.rec_refpos[rec + 1] <= (uint64_t)toff) rec++;      // last record starting at or before toff
	const uint64_t p = r.rec_refpos[rec];
	if ((uint64_t)toff < p || (uint64_t)toff >= p + r.rec_len[rec]) return 4;
	const uint64_t bo = r.rec_bufpos[rec] + ((uint64_t)toff - p);
	return (r.buf[bo >> 2] >> ((bo & 3) << 1)) & 3;
}

} // namespace bt2g
#endif
