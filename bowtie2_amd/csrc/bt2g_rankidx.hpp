// bt2g_rankidx.hpp -- from the on-disk index sections to the HBM layout of bt2g_device.hpp.
//
// The .1.bt2[l] file keeps the BWT as "sides" (bt2_idx.h:133-167, writer :2829-3174): sideBwtSz = 48 / 96 bytes of 2-bit
// characters (character i of a side at bits 2*(i&3) of byte i>>2) followed by occ[A,C,G,T] = occurrences BEFORE the side ('$' is
// stored as A and not counted).  The .2.bt2[l] file keeps the suffix array of every 2^offRate-th row.  Neither is a shape a GPU
// lane can query cheaply (see bt2g_device.hpp), and the on-disk format is fixed, so bt2g_index_load transcodes once:
//   make_rank_block   one 64-byte RankBlock per 128 BWT characters
//   sa_segment        the full suffix array, by the reference's own walk (Ebwt::getOffset, bt2_idx.cpp:150-171), organised so
//                     that every row is visited once: the rows between two consecutive SA samples along the LF chain form a
//                     segment; the walk from the sample that heads a segment reaches the next sample after m steps, and row k
//                     of the segment then resolves to offs[next sample] + (m - k) after (m - k) steps -- exactly what the
//                     reference computes for that row.  2 LF steps per row in total (one pass to find m, one to write).
// The functions are written once for host and device: the kernels of bt2g_rankidx.hip call them per lane, the host loops at the
// bottom serve the CPU test twin (tests/hostsim), which has no device.
#ifndef BT2G_RANKIDX_HPP_
#define BT2G_RANKIDX_HPP_

#include "bt2g_device.hpp"

namespace bt2g {

// character at BWT position p of the verbatim sides (A past the end of the last side)
template <typename TOff>
BT2_HD int side_bwt_char(const uint8_t* ebwt, uint64_t n_sides, uint64_t p) {
	const uint64_t s = p / OffTraits<TOff>::kSideBwtLen;
	if (s >= n_sides) return 0;
	const uint32_t o = (uint32_t)(p % OffTraits<TOff>::kSideBwtLen);
	return (ebwt[s * OffTraits<TOff>::kSideSz + (o >> 2)] >> ((o & 3) << 1)) & 3;
}

// Block b of the rank index.  fchr = Ebwt::fchr, zoff = row of '$'.
template <typename TOff>
BT2_HD void make_rank_block(const uint8_t* ebwt, uint64_t n_sides, uint64_t b, const TOff fchr[5], TOff zoff, RankBlock& out) {
	constexpr uint32_t L = OffTraits<TOff>::kSideBwtLen, SZ = OffTraits<TOff>::kSideSz, BWT_SZ = L / 4;
	const uint64_t P = b << kBlkShift;
	uint64_t s = P / L;
	uint32_t coff = (uint32_t)(P % L);
	if (s >= n_sides) { s = n_sides - 1; coff = L; }      // the block past the last side: totals of the whole BWT
	uint64_t occ[4];
	const uint8_t* side = ebwt + s * SZ;
	for (int c = 0; c < 4; c++) {
		TOff v;
		const uint8_t* q = side + BWT_SZ + (uint32_t)c * (uint32_t)sizeof(TOff);
		v = 0;
		for (uint32_t k = 0; k < sizeof(TOff); k++) v |= (TOff)q[k] << (8 * k);
		occ[c] = (uint64_t)v + (uint64_t)fchr[c];
	}
	for (uint32_t i = 0; i < coff; i++) occ[(side[i >> 2] >> ((i & 3) << 1)) & 3]++;
	const uint64_t side0 = s * L;
	if ((uint64_t)zoff >= side0 && (uint64_t)zoff < side0 + coff) occ[0]--;      // the '$' row was tallied as an A
	for (int c = 0; c < 4; c++) out.occ[c] = occ[c];
	for (uint32_t w = 0; w < 4; w++) {
		uint32_t p0 = 0, p1 = 0;
		for (uint32_t i = 0; i < 32; i++) {
			const int ch = side_bwt_char<TOff>(ebwt, n_sides, P + 32 * w + i);
			p0 |= (uint32_t)(ch & 1) << i;
			p1 |= (uint32_t)(ch >> 1) << i;
		}
		out.p0[w] = p0; out.p1[w] = p1;
	}
}
BT2_HD uint64_t rank_block_count(uint64_t n_sides, uint32_t side_bwt_len) { return ((n_sides * side_bwt_len) >> kBlkShift) + 2; }

// ---- full suffix array ----
template <typename TOff>
BT2_HD bool sa_sampled(const DevEbwt<TOff>& e, TOff row) { return (row & (TOff)(((TOff)OffTraits<TOff>::kMask) << e.off_rate)) == row; }

// entries every walk ends at: SA samples (0 steps) and the '$' row (offset 0 -- getOffset tests for it first)
template <typename TOff>
BT2_HD void sa_init_row(const DevEbwt<TOff>& e, const TOff* offs, TOff row, uint64_t* sa) {
	if (row == e.zoff) sa[(uint64_t)row] = joff_pack(0, 0);
	else if (sa_sampled(e, row)) sa[(uint64_t)row] = joff_pack((uint64_t)offs[(uint64_t)row >> e.off_rate], 0);
}

// The segment headed by r0 (an SA sample, or row `len`: the row of the empty suffix heads the LF chain).  Returns its length.
// The reference samples the suffix array by ROW ((row & mask) == row, bt2_idx.h), so a segment's length is geometric with mean 2^offRate and
// has no upper bound: a segment longer than the 16-bit step count next to the offset holds (joff_pack) cannot be represented -- the rows past
// that point would read as "no offset" and their seed hits would be dropped silently.  Such rows are counted (`lost`) and the index load
// fails when there are any (bt2g_index_load; only sparse samples, --offrate 12 and up on a genome-sized text, get there at all).
template <typename TOff>
BT2_HD uint32_t sa_segment(const DevEbwt<TOff>& e, const TOff* offs, TOff r0, uint64_t* sa, uint32_t* lost = nullptr) {
	if (r0 == e.zoff) return 0;
	TOff r = r0;
	uint32_t m = 0;
	for (;;) {
		map_lf1(e, r);
		m++;
		if (r == e.zoff || sa_sampled(e, r)) break;
	}
	const uint64_t base = r == e.zoff ? 0ull : (uint64_t)offs[(uint64_t)r >> e.off_rate];
	if (!sa_sampled(e, r0)) sa[(uint64_t)r0] = joff_pack(base + m, m);      // (row len when it is not a sample itself)
	// rows stored with a step count: 1 .. m-1 when the head is itself a sample (it keeps its own offset), 1 .. m otherwise
	if (lost) { const uint32_t m_eff = sa_sampled(e, r0) ? m - 1 : m; if (m_eff >= 0xffffu) *lost += m_eff - 0xfffeu; }
	r = r0;
	for (uint32_t k = 1; k < m; k++) {
		map_lf1(e, r);
		sa[(uint64_t)r] = joff_pack(base + (m - k), m - k);
	}
	return m;
}
// number of segment heads: samples 0, S, 2S, ... <= len, plus row len
template <typename TOff>
BT2_HD uint64_t sa_n_heads(const DevEbwt<TOff>& e) { return ((uint64_t)e.len >> e.off_rate) + 2; }
template <typename TOff>
BT2_HD TOff sa_head_row(const DevEbwt<TOff>& e, uint64_t h) {
	const uint64_t ns = ((uint64_t)e.len >> e.off_rate) + 1;       // samples
	if (h < ns) return (TOff)(h << e.off_rate);
	return sa_sampled(e, e.len) ? e.zoff : e.len;                  // (zoff = "nothing to do")
}

#if !defined(__HIP_DEVICE_COMPILE__)
// host loops (CPU test twin only)
template <typename TOff>
inline void host_make_rank_blocks(const uint8_t* ebwt, uint64_t n_sides, const TOff fchr[5], TOff zoff, RankBlock* out, uint64_t n_blocks) {
	for (uint64_t b = 0; b < n_blocks; b++) make_rank_block<TOff>(ebwt, n_sides, b, fchr, zoff, out[b]);
}
template <typename TOff>
inline void host_make_full_sa(const DevEbwt<TOff>& e, const TOff* offs, uint64_t* sa) {
	for (uint64_t r = 0; r <= (uint64_t)e.len; r++) sa_init_row(e, offs, (TOff)r, sa);
	const uint64_t nh = sa_n_heads(e);
	for (uint64_t h = 0; h < nh; h++) sa_segment(e, offs, sa_head_row(e, h), sa);
}
template <typename TOff>
inline uint64_t host_make_full_sa_checked(const DevEbwt<TOff>& e, const TOff* offs, uint64_t* sa) {
	for (uint64_t r = 0; r <= (uint64_t)e.len; r++) sa_init_row(e, offs, (TOff)r, sa);
	const uint64_t nh = sa_n_heads(e);
	uint64_t lost = 0;
	for (uint64_t h = 0; h < nh; h++) { uint32_t l = 0; sa_segment(e, offs, sa_head_row(e, h), sa, &l); lost += l; }
	return lost;
}
#endif

} // namespace bt2g
#endif
