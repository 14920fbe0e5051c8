// bt2g_device.hpp -- device-side view of the FM index and the rank/LF primitives.
//
// HBM layout (built once per bt2g_index_load from the verbatim .1.bt2[l] / .2.bt2[l] sections, bt2g_rankidx.hpp):
//  * the BWT as RANK BLOCKS: one aligned 64-byte line per 128 characters = absolute occ[A,C,G,T] (64-bit, fchr folded in, the '$'
//    row already discounted) + the characters as two bit planes of 4 x 32 bits.  One rank query is one line read; matching a
//    character is two XORs and an AND per 32 positions, counting one v_bcnt per word; row -> block is a shift.  The on-disk side
//    (48/96 bytes of 2-bit pairs + occ per 192/384 rows, bt2_idx.h:133-167) costs ~8x the vector instructions per query (12 x 64-bit
//    words to mask per character, division by 384), which is what bound the FM kernels in round 2 (profiles/r02z_pmc_*).
//  * the suffix array in FULL: sa[row] = joined-text offset of the row | (LF steps the reference's walk to its SA sample takes) << 48.
//    The file keeps every 2^offRate-th row (Ebwt::getOffset walks LF to the next sampled row, bt2_idx.cpp:150-171: 16 dependent
//    side reads on average); 288 GB of HBM hold the whole array (8 bytes x 3.1 G rows = 25 GB for hg38), so resolving a row is ONE
//    8-byte read.  The step count is kept because the path's work counters and the roofline accounting (one side per
//    LF step) are defined by the reference's walk.
// ftab/eftab/rstarts/plen are uploaded verbatim at index width TOff (uint32_t for .bt2, uint64_t for .bt2l).
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
#define BT2_HDI __host__ __device__ __forceinline__   // large functions with ONE call site in a kernel body: inlined, so that no prologue saves callee-saved registers
#define BT2_D __device__ __forceinline__
#else
#define BT2_HD inline
#define BT2_HDN
#define BT2_HDI
#define BT2_D inline
#endif

#if !defined(__HIPCC__)
struct ulonglong2 { unsigned long long x, y; };   // host-side stand-ins (test builds only)
struct uint4 { unsigned int x, y, z, w; };
#endif

// BT2_G qualifies a type as living in device global memory (HBM).  An access through a BT2_G pointer or reference is a global_*
// instruction; through a plain (generic) one it is a flat_* instruction, and -- worse for a worker whose control code is wave-uniform --
// the compiler must treat every value loaded through a generic pointer as lane-varying (the address could be per-lane scratch), so that
// everything computed from it runs on the vector ALU and every branch on it becomes exec-mask control flow.  The worker therefore
// reaches its work area, its alignment records and its result records through BT2_G types.  (Only the device compilation knows address
// spaces; for the host pass and the CPU test twin the qualifier is empty.)
#if defined(__HIP_DEVICE_COMPILE__)
#define BT2_G __attribute__((address_space(1)))
#else
#define BT2_G
#endif
// a BT2_G pointer as the generic pointer the lane-parallel platform helpers take (they name the address space again per access)
template <typename T> BT2_HD T* gen_ptr(BT2_G T* p) { return (T*)p; }

// Scalar loads / stores for pointers that are KNOWN to point into the wave's arena in HBM.  Through a plain (generic)
// pointer the compiler must issue FLAT instructions: it then has to assume the access may alias LDS (so every LDS value it
// holds in registers is reloaded after a store) and the wait for a FLAT load also drains the LDS counter.  The device
// version names the global address space; the host twin (tests/hostsim) is an ordinary dereference.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> __host__ __device__ __forceinline__ T gld(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }
template <typename T> __host__ __device__ __forceinline__ void gst(T* p, T v) { *(__attribute__((address_space(1))) T*)p = v; }
template <typename T> __host__ __device__ __forceinline__ T gld(const BT2_G T* p) { return *p; }        // (pointers already typed as global)
template <typename T> __host__ __device__ __forceinline__ void gst(BT2_G T* p, T v) { *p = v; }
#else
template <typename T> BT2_HD T gld(const T* p) { return *p; }
template <typename T> BT2_HD void gst(T* p, T v) { *p = v; }
#endif

namespace bt2g {

// geometry of the on-disk sides: what one rank query costs in the path's accounting
template <typename TOff> struct OffTraits;
template <> struct OffTraits<uint32_t> {
	static constexpr uint32_t kSideSz = 64, kSideBwtLen = 192, kBwtWords = 6; // 48 B of BWT = 6 x u64
	static constexpr uint32_t kMask = 0xffffffffu;
	static constexpr uint32_t kSideShift = 6;    // kSideBwtLen = 3 << kSideShift
};
template <> struct OffTraits<uint64_t> {
	static constexpr uint32_t kSideSz = 128, kSideBwtLen = 384, kBwtWords = 12; // 96 B of BWT = 12 x u64
	static constexpr uint64_t kMask = ~0ull;
	static constexpr uint32_t kSideShift = 7;
};

// One rank block: 128 BWT characters.  Character i of the block has bit (i & 31) of p0[i >> 5] = low bit of its code and of
// p1[i >> 5] = high bit.  occ[c] = fchr[c] + (# of c in the BWT before this block), the '$' row (stored as A in the file,
// bt2_idx.h:1766-1774) not counted.
struct alignas(64) RankBlock { uint64_t occ[4]; uint32_t p0[4], p1[4]; };
constexpr uint32_t kBlkShift = 7, kBlkLen = 128;

// full suffix array entry: joined offset in the low 48 bits, LF steps of the reference's walk in the high 16
constexpr uint64_t kJoffNone = ~0ull;
BT2_HD uint64_t joff_pack(uint64_t joff, uint32_t steps) { return (joff >> 48) == 0 && steps < 0xffffu ? (joff | ((uint64_t)steps << 48)) : kJoffNone; }

template <typename TOff>
struct DevEbwt {
	const RankBlock* blk;  // rank blocks
	const TOff*    ftab;
	const TOff*    eftab;
	const uint64_t* sa;    // full suffix array (forward index only), joff_pack format
	TOff len, zoff;
	uint64_t zblk;         // block and position within it of the '$' row
	uint32_t zchar;
	uint32_t ftab_chars, off_rate;
	uint32_t is_fw;
	TOff fchr[5];
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

// Work counters of the stage kernels (roofline accounting).  The device holds kCntSlots copies, one cache line each, and a wave adds to the
// copy its workgroup index selects: with ONE copy every wave of a launch sent its atomics to the same three words of one line, and the L2
// channel that owns the line serialised them -- 1.9 M atomics per launch of 40 M lanes, ~14 ns each: 27 of the 43 ms k_extend_hits took in
// rounds 1-3, and half of k_seed_search_exact, were this.  The host sums the copies when it reads them.
struct alignas(64) DevCounters {
	unsigned long long rank_queries, sa_lookups, ftab_lookups, dp_cells, bwops;
};
constexpr unsigned kCntSlots = 256;

// ---------------------------------------------------------------------------------------
BT2_HD int popc32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
	return __popc(x);
#else
	return __builtin_popcount(x);
#endif
}

// One block, loaded into registers: four 16-byte loads of one aligned 64-byte line.  The planes and the counts are VECTOR values on the
// device (ext_vector_type): a rank query picks a plane word by the row's position and a count by the character, both run-time indices,
// and the optimiser turns any such pick from a plain array -- even one written as a chain of compares -- into an indexed load, which
// puts the whole block into scratch memory (64 bytes stored and re-read per lane and rank query; every FM kernel of rounds 1-3 did).
// An element picked from a vector value is a register select.
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t blk_u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t blk_u64x4 __attribute__((ext_vector_type(4)));
#else
struct blk_u32x4 { uint32_t v[4]; BT2_HD uint32_t& operator[](uint32_t i) { return v[i]; } BT2_HD const uint32_t& operator[](uint32_t i) const { return v[i]; } };
struct blk_u64x4 { uint64_t v[4]; BT2_HD uint64_t& operator[](uint32_t i) { return v[i]; } BT2_HD const uint64_t& operator[](uint32_t i) const { return v[i]; } };
#endif
struct Blk { blk_u64x4 occ; blk_u32x4 p0, p1; };

BT2_HD void load_blk(const RankBlock* blks, uint64_t b, Blk& o) {
	const uint4* p = reinterpret_cast<const uint4*>(blks + b);
	const uint4 a = p[0], c = p[1], d = p[2], e = p[3];
	o.occ[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); o.occ[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
	o.occ[2] = (uint64_t)c.x | ((uint64_t)c.y << 32); o.occ[3] = (uint64_t)c.z | ((uint64_t)c.w << 32);
	o.p0[0] = d.x; o.p0[1] = d.y; o.p0[2] = d.z; o.p0[3] = d.w;
	o.p1[0] = e.x; o.p1[1] = e.y; o.p1[2] = e.z; o.p1[3] = e.w;
}

// bits 0..n-1 of word w of a block prefix of `off` characters (off in 0..128)
BT2_HD uint32_t prefix_mask(uint32_t off, uint32_t w) {
	const int n = (int)off - (int)(32 * w);
	const uint32_t nn = n < 0 ? 0u : (n > 32 ? 32u : (uint32_t)n);
	return ~(uint32_t)(~0ull << nn);
}

// # of chars == c among the first `off` chars of the block
BT2_HD uint32_t blk_count(const Blk& s, uint32_t off, int c) {
	const uint32_t k0 = (c & 1) ? 0u : ~0u, k1 = (c & 2) ? 0u : ~0u;
	uint32_t cnt = 0;
#pragma unroll
	for (uint32_t w = 0; w < 4; w++) cnt += (uint32_t)popc32((s.p0[w] ^ k0) & (s.p1[w] ^ k1) & prefix_mask(off, w));
	return cnt;
}

BT2_HD int blk_char(const Blk& s, uint32_t off) {
	const uint32_t w = off >> 5;
	const uint32_t a = s.p0[w], b = s.p1[w];
	const uint32_t sh = off & 31;
	return (int)(((a >> sh) & 1u) | (((b >> sh) & 1u) << 1));
}

// rank_c(row) = fchr[c] + occ_side[c] + count - ['$' fix]   (countBt2Side, bt2_idx.h:1758)
template <typename TOff>
BT2_HD TOff rank_in_blk(const DevEbwt<TOff>& e, const Blk& s, uint64_t b, uint32_t off, int c) {
	uint32_t cnt = blk_count(s, off, c);
	if (c == 0 && b == e.zblk && off > e.zchar) cnt--;   // '$' is stored as 'A' (bt2_idx.h:1766-1774)
	return (TOff)(s.occ[c] + cnt);
}

// all four characters at once (countBt2SideEx, bt2_idx.h:1887)
template <typename TOff>
BT2_HD void rank4_in_blk(const DevEbwt<TOff>& e, const Blk& s, uint64_t b, uint32_t off, TOff out[4]) {
	uint32_t c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
	for (uint32_t w = 0; w < 4; w++) {
		const uint32_t m = prefix_mask(off, w), a = s.p0[w], h = s.p1[w];
		c1 += (uint32_t)popc32(a & ~h & m);
		c2 += (uint32_t)popc32(~a & h & m);
		c3 += (uint32_t)popc32(a & h & m);
	}
	uint32_t c0 = off - c1 - c2 - c3;
	if (b == e.zblk && off > e.zchar) c0--;
	out[0] = (TOff)(s.occ[0] + c0);
	out[1] = (TOff)(s.occ[1] + c1);
	out[2] = (TOff)(s.occ[2] + c2);
	out[3] = (TOff)(s.occ[3] + c3);
}

// the side of the on-disk layout a row falls into: what the accounting of "sides read" is defined on (one rank query = one side,
// two loci in one side = one read: SideLocus::initFromTopBot, bt2_idx.h:325-348)
template <typename TOff>
BT2_HD uint32_t side_of(TOff row) { return (uint32_t)(((uint64_t)row >> OffTraits<TOff>::kSideShift) / 3u); }

template <typename TOff>
BT2_HD TOff rank1(const DevEbwt<TOff>& e, TOff row, int c) {
	const uint64_t b = (uint64_t)row >> kBlkShift;
	Blk s;
	load_blk(e.blk, b, s);
	return rank_in_blk(e, s, b, (uint32_t)row & (kBlkLen - 1), c);
}

template <typename TOff>
BT2_HD void rank4(const DevEbwt<TOff>& e, TOff row, TOff out[4]) {
	const uint64_t b = (uint64_t)row >> kBlkShift;
	Blk s;
	load_blk(e.blk, b, s);
	rank4_in_blk(e, s, b, (uint32_t)row & (kBlkLen - 1), out);
}

// Pair query rank_c(top), rank_c(bot).  Returns the # of sides the reference reads for it (1 when both loci share a side).
template <typename TOff>
BT2_HD int rank1_pair(const DevEbwt<TOff>& e, TOff top, TOff bot, int c, TOff& otop, TOff& obot) {
	const uint64_t bt = (uint64_t)top >> kBlkShift, bb = (uint64_t)bot >> kBlkShift;
	Blk s;
	load_blk(e.blk, bt, s);
	otop = rank_in_blk(e, s, bt, (uint32_t)top & (kBlkLen - 1), c);
	if (bb != bt) load_blk(e.blk, bb, s);
	obot = rank_in_blk(e, s, bb, (uint32_t)bot & (kBlkLen - 1), c);
	return side_of(top) == side_of(bot) ? 1 : 2;
}

template <typename TOff>
BT2_HD int rank4_pair(const DevEbwt<TOff>& e, TOff top, TOff bot, TOff t[4], TOff b[4]) {
	const uint64_t bt = (uint64_t)top >> kBlkShift, bb = (uint64_t)bot >> kBlkShift;
	Blk s;
	load_blk(e.blk, bt, s);
	rank4_in_blk(e, s, bt, (uint32_t)top & (kBlkLen - 1), t);
	if (bb != bt) load_blk(e.blk, bb, s);
	rank4_in_blk(e, s, bb, (uint32_t)bot & (kBlkLen - 1), b);
	return side_of(top) == side_of(bot) ? 1 : 2;
}

// mapLF1(row, l, c) (bt2_idx.h:2420): all-ones if BWT[row] != c or row is the '$' row
template <typename TOff>
BT2_HD TOff map_lf1c(const DevEbwt<TOff>& e, TOff row, int c) {
	const uint64_t b = (uint64_t)row >> kBlkShift;
	const uint32_t off = (uint32_t)row & (kBlkLen - 1);
	Blk s;
	load_blk(e.blk, b, s);
	if (blk_char(s, off) != c || row == e.zoff) return (TOff)OffTraits<TOff>::kMask;
	return rank_in_blk(e, s, b, off, c);
}

// mapLF1(row&, l) (bt2_idx.h:2451): returns BWT char (or -1 at '$') and advances row
template <typename TOff>
BT2_HD int map_lf1(const DevEbwt<TOff>& e, TOff& row) {
	if (row == e.zoff) return -1;
	const uint64_t b = (uint64_t)row >> kBlkShift;
	const uint32_t off = (uint32_t)row & (kBlkLen - 1);
	Blk s;
	load_blk(e.blk, b, s);
	const int c = blk_char(s, off);
	row = rank_in_blk(e, s, b, off, c);
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

// Ebwt::getOffset (bt2_idx.cpp:150): the joined-text offset of a row and the number of LF steps the reference's walk to its
// SA sample takes -- one read of the full suffix array (bt2g_rankidx.hpp derives it from the sample with that very walk).
template <typename TOff>
BT2_HD TOff get_offset(const DevEbwt<TOff>& e, TOff row, uint32_t& nsteps) {
	const uint64_t v = e.sa[(uint64_t)row];
	nsteps = (uint32_t)(v >> 48);
	return (TOff)(v & 0xffffffffffffull);
}

// Ebwt::joinedToTextOff (bt2_idx.cpp:54) for the forward index.  tidx = all-ones if rejected.
template <typename TOff>
BT2_HD void joined_to_text_off(const DevIndex<TOff>& ix, TOff qlen, TOff off, TOff& tidx, TOff& textoff, TOff& tlen,
                               bool reject_straddle, bool& straddled, uint64_t* frag = nullptr) {      // frag: -> {joined start, length, text offset, reference} of the fragment
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
				if (frag) { frag[0] = (uint64_t)lower; frag[1] = (uint64_t)(upper - lower); frag[2] = (uint64_t)ix.rstarts[(uint64_t)elt * 3 + 2]; frag[3] = (uint64_t)tidx; }
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
