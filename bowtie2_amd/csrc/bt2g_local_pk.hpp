// bt2g_local_pk.hpp -- the cell arithmetic of the local-mode DP fill, two cells per 32-bit register (16 bits each).
//
// Local scores are small non-negative numbers (floor 0, at most read length x match bonus), so a cell's H / E / F fit 16 bits and gfx950's
// packed 16-bit ALU (v_pk_add_i16, v_pk_max_i16, v_pk_sub_u16 clamp, v_pk_min_u16, v_pk_mad_i16) computes two cells per instruction.  The two
// cells of a register are never neighbours -- cells on one anti-diagonal are the independent ones -- so a lane owns TWO blocks of RB consecutive
// rows: block `lane` in the low halves and block `lane + 64` in the high halves; block k works on column t - k in step t (dp_cell_pk,
// bt2g_align.hpp).  What the reference computes per cell (alignNucleotidesLocalSseU8, aligner_swsse_loc_u8.cpp:240-700; the 16-bit kernel agrees
// wherever the 8-bit one does not saturate) is restated here once, for the device fill (bt2g_align_kernel.hip: fill_local_pk) and for the check
// the CPU twin runs on every local window it fills (tests/hostsim/hostsim.cpp, BT2G_CHECK_LOCAL_PK=1): same source, packed operations spelled as
// builtins on the device and as two scalar halves on the host.
//
// No selects in the row loop: everything conditional is a constant mask prepared per row (gap veto, row validity) or per step (reference N), and
// a block that has not reached column 0 yet computes on zeros fed with "reference N" and stays zero, so nothing needs an `active` guard except
// the stores.  The predecessor byte (PB_*) is assembled from "not equal" flags  min(x ^ y, 1)  and inverted once.
#pragma once
#include <cstdint>
#include "bt2g_device.hpp"

namespace bt2g {
namespace pk {
#if defined(__HIP_DEVICE_COMPILE__)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s(uint32_t a) { return __builtin_bit_cast(s16x2, a); }
__device__ __forceinline__ u16x2 as_u(uint32_t a) { return __builtin_bit_cast(u16x2, a); }
__device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, as_s(a) + as_s(b)); }
__device__ __forceinline__ uint32_t maxs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ uint32_t maxu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(as_u(a), as_u(b))); }
__device__ __forceinline__ uint32_t minu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(as_u(a), as_u(b))); }
__device__ __forceinline__ uint32_t subsu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(as_u(a), as_u(b))); }
__device__ __forceinline__ uint32_t mad(uint32_t a, uint32_t b, uint32_t c) { return __builtin_bit_cast(uint32_t, as_s(a) * as_s(b) + as_s(c)); }
// per half: 1 where a != 0.  Spelled as the instruction: written as min(a, 1) the compiler recognises "a != 0", splits the halves and
// rebuilds the flag with a compare, a select and a byte permute per half -- five instructions for one.
__device__ __forceinline__ uint32_t nz(uint32_t a) { uint32_t r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(0x00010001u)); return r; }
// (a << 1) | b, as the one instruction it is (left to itself the compiler re-associates a chain of these into shifts by 1 .. 6 and 3-input ors: half again as many)
__device__ __forceinline__ uint32_t shl1_or(uint32_t a, uint32_t b) { uint32_t r; asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
inline uint32_t shl1_or(uint32_t a, uint32_t b) { return (a << 1) | b; }
inline uint32_t mk(uint32_t lo, uint32_t hi) { return (lo & 0xffffu) | (hi << 16); }
inline uint32_t add(uint32_t a, uint32_t b) { return mk(a + b, (a >> 16) + (b >> 16)); }
inline uint32_t maxs(uint32_t a, uint32_t b) {
	const int16_t al = (int16_t)a, ah = (int16_t)(a >> 16), bl = (int16_t)b, bh = (int16_t)(b >> 16);
	return mk((uint16_t)(al > bl ? al : bl), (uint16_t)(ah > bh ? ah : bh));
}
inline uint32_t maxu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16; return mk(al > bl ? al : bl, ah > bh ? ah : bh); }
inline uint32_t minu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16; return mk(al < bl ? al : bl, ah < bh ? ah : bh); }
inline uint32_t subsu(uint32_t a, uint32_t b) { const uint32_t al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16; return mk(al > bl ? al - bl : 0u, ah > bh ? ah - bh : 0u); }
inline uint32_t mad(uint32_t a, uint32_t b, uint32_t c) { return mk((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu), (a >> 16) * (b >> 16) + (c >> 16)); }
inline uint32_t nz(uint32_t a) { return minu(a, 0x00010001u); }
#endif
BT2_HD uint32_t both(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }      // the same 16-bit value in both halves
BT2_HD uint32_t ne(uint32_t a, uint32_t b) { return nz(a ^ b); }                  // per half: 1 where a != b
}  // namespace pk

// per-row constants of one lane (row r of its low-half block in the low 16 bits, of its high-half block in the high 16 bits)
template <int RB> struct LocalPkRows {
	uint32_t rowmask[RB];   // 1 << read character (16 for N and for rows beyond the read)
	uint32_t bpm[RB];       // match bonus + mismatch penalty: score = -penalty + match * (bonus + penalty)
	uint32_t nmmp[RB];      // -mismatch penalty (read N: -N penalty)
	uint32_t vm[RB];        // 0xffff where the row may open / extend gaps (not within gapbar of either end, and a row of the read)
	uint32_t okm[RB];       // 0xffff where the row is a row of the read
};
struct LocalPkPen { uint32_t rdgapo, rdgape, rfgapo, rfgape, npen_neg; };      // both halves each
BT2_HD LocalPkPen local_pk_pen(int rdgapo, int rdgape, int rfgapo, int rfgape, int n_pen) {
	LocalPkPen p; p.rdgapo = pk::both(rdgapo); p.rdgape = pk::both(rdgape); p.rfgapo = pk::both(rfgapo); p.rfgape = pk::both(rfgape); p.npen_neg = pk::both(-n_pen);
	return p;
}
// constants of one row (one half): OR the results of the two halves, the high one shifted by 16
struct LocalPkRow1 { uint32_t rowmask, bpm, nmmp, vm, okm; };
BT2_HD LocalPkRow1 local_pk_row(bool valid, int rdc, int mmp, bool veto, int match_bonus, int n_pen) {
	LocalPkRow1 c;
	c.rowmask = 1u << (valid ? rdc : 4);
	const int pen = (valid && rdc <= 3) ? mmp : n_pen;
	c.bpm = (uint32_t)(match_bonus + pen) & 0xffffu; c.nmmp = (uint32_t)(-pen) & 0xffffu;
	c.vm = (valid && !veto) ? 0xffffu : 0u; c.okm = valid ? 0xffffu : 0u;
	return c;
}

// One step of one lane: its RB rows against the reference character(s) `refm` (mask per half, 16 = N), given the cells above its blocks:
// hdiag = H(row above, column - 1), fin_h / fin_f = H / F(row above, this column), cm = column maximum so far.  Hp / Ep are the lane's H / E of the
// previous column and become this column's; pb[r] = predecessor bytes (one per half, bits 0-6 and 16-22); hlast / flast = H / F of the lane's
// last row; cm = column maximum including these rows.  Bits of pb that nobody may look at (E bits of a cell whose E is 0, ... -- see the fill's
// comment in bt2g_align_kernel.hip) are whatever the arithmetic leaves there.
template <int RB>
BT2_HD void local_pk_step(const LocalPkRows<RB>& K, const LocalPkPen& P, uint32_t refm, uint32_t hdiag, uint32_t fin_h, uint32_t fin_f,
                          uint32_t (&Hp)[RB], uint32_t (&Ep)[RB], uint32_t (&pb)[RB], uint32_t& hlast, uint32_t& flast, uint32_t& cm) {
	// reference N (no A/C/G/T bit): every row scores -N penalty
	const uint32_t nm = pk::add(pk::nz(refm & 0x000f000fu), 0xffffffffu);      // 0xffff where N
#pragma unroll
	for (int r = 0; r < RB; r++) {
		const uint32_t mt = pk::nz(refm & K.rowmask[r]);      // 1 where the characters match (N row against N reference too: overridden below)
		uint32_t sc = pk::mad(mt, K.bpm[r], K.nmmp[r]);
		sc = (P.npen_neg & nm) | (sc & ~nm);
		const uint32_t eo = pk::subsu(Hp[r], P.rdgapo) & K.vm[r], ee = pk::subsu(Ep[r], P.rdgape);
		const uint32_t e = pk::maxu(eo, ee);
		const uint32_t fo = pk::subsu(fin_h, P.rfgapo), fe = pk::subsu(fin_f, P.rfgape);
		const uint32_t f = pk::maxu(fo, fe) & K.vm[r];
		const uint32_t hd = pk::add(hdiag, sc);
		const uint32_t h = pk::maxs(pk::maxs(hd, e), f);
		// "not a predecessor" flags, one bit per half each, PB_FE first (shift-and-or per flag)
		uint32_t n = pk::ne(fe, f);                                                // PB_FE
		n = pk::shl1_or(n, pk::ne(fo, f));                                             // PB_FO
		n = pk::shl1_or(n, pk::ne(ee, e));                                             // PB_EE
		n = pk::shl1_or(n, pk::ne(eo, e));                                             // PB_EO
		n = pk::shl1_or(n, pk::ne(h, f));                                             // PB_HF
		n = pk::shl1_or(n, pk::ne(h, e));                                             // PB_HE
		n = pk::shl1_or(n, pk::ne(hd, h) | (pk::nz(hdiag) ^ 0x00010001u));             // PB_HD: the diagonal, and only if it is above the floor
		pb[r] = n ^ 0x007f007fu;
		hdiag = Hp[r];
		Hp[r] = h; Ep[r] = e;
		fin_h = h; fin_f = f;
		cm = pk::maxu(cm, h & K.okm[r]);
	}
	hlast = fin_h; flast = fin_f;
}

}  // namespace bt2g
