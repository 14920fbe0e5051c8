// bt2g_align_kernel.hip -- the per-read multiseed worker as one gfx950 kernel.
//
// One wavefront per read (see bt2g_align.hpp).  Waves are persistent: each pulls the next read
// index from a device counter, so long reads / repetitive reads do not leave a tail of idle
// CUs.  Per-wave working state (Work + DP scratch) lives in an HBM arena sized at launch.
//
// The wave-parallel pieces (DevPlat): the DP fills -- the end-to-end 8-bit fill over the band of diagonals an alignment can touch (lane = 2 RP
// diagonals, packed 16-bit ALU, one predecessor byte per cell), the 16-bit end-to-end fill on the anti-diagonal wavefront, the local fill
// (two blocks of rows per lane, packed, predecessor bytes in anti-diagonal order: bt2g_local_pk.hpp) -- each as leaf functions; the tile
// fetches and diagonal / gap runs of the backtrace; the candidate gather and its radix sort; the row sampler's register table.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <type_traits>
#include "bt2g_align_core.hpp"
#include "bt2g_local_pk.hpp"
#include "bt2g_align_kernel.hpp"

namespace bt2g {

__shared__ HotWork g_hot;    // one wavefront per workgroup: the hot per-read state lives in LDS
// control blocks of the launch / of the read in flight, also in LDS and also reached by name (DevPlat::params() ...): see the
// note at the PRM / RPR / IX / PRE macros in bt2g_align_core.hpp
__shared__ AlignParams g_P;
__shared__ ReadParams g_rp;
__shared__ PreComp g_pre;
__shared__ AlState g_st;     // the worker's own state (Aligner has no data members)
// The per-column tail of the hot state (hot_tail_bytes, bt2g_align.hpp): dynamic LDS, sized by the launch from the widest DP window it must hold.
extern __shared__ __attribute__((aligned(16))) uint8_t g_tail[];
// Where the dynamic part starts depends on the kernel (its static LDS), so a function that is really called finds `g_tail` through a table in
// memory (llvm.amdgcn.dynlds.offset.table): a vector-memory load -- and a wait on the vector-memory counter, which also waits for every store
// in flight -- at each use.  The kernels, which know the address, leave the LDS addresses of the tail's parts in g_st (static LDS) instead.
typedef __attribute__((address_space(3))) uint8_t lds_byte;
__device__ __forceinline__ uint32_t lds_addr_of_tail() { return (uint32_t)reinterpret_cast<uintptr_t>((lds_byte*)g_tail); }
__device__ __forceinline__ uint8_t* lds_at(uint32_t a) { return (uint8_t*)reinterpret_cast<lds_byte*>((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a)); }
__device__ __forceinline__ uint8_t* dev_rf() { return lds_at(g_st.rf_at); }                      // reference masks of the current DP window
__device__ __forceinline__ int16_t* dev_lastrow() { return reinterpret_cast<int16_t*>(lds_at(g_st.ned_at)); }      // scores of the last DP row, clamped at -32768 (gatherCells)
__device__ __forceinline__ Edit* dev_ned() { return reinterpret_cast<Edit*>(lds_at(g_st.ned_at)); }                 // edits of the backtrace in progress
// Behind the tail, when the launch has LDS to spare (launch_align): the reportedThrough plane of the end-to-end band matrix in hand, ONE BIT per cell
// (bit  row * w + diagonal,  w = the band's row width): the walks mark and test cells without a store to drain or a word to fetch.
// (Session r05c also kept the predecessor bytes of the matrix's last rows there, for the 23 of 24 backtrace attempts per read that fail within
// a few cells of the last row: 351 -> 379 ms per 2 M reads at the same occupancy -- those walks are not waiting for memory.  Dropped.)
__device__ __forceinline__ uint32_t* dev_rt() { return reinterpret_cast<uint32_t*>(lds_at(g_st.rt_at)); }
alignas(16) __shared__ unsigned char g_ix_raw[sizeof(DevIndex<uint64_t>) > sizeof(DevIndex<uint32_t>) ? sizeof(DevIndex<uint64_t>) : sizeof(DevIndex<uint32_t>)];

// Memory written by some lanes of the wave and read by others afterwards.  (One workgroup = one wavefront: the compiler knows the largest
// workgroup is 64 threads and lowers the workgroup-scope fence to a wavefront-scope one -- no s_waitcnt, just a barrier for its own reordering.)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// ---------------------------------------------------------------------------------------------------------------------
// End-to-end 8-bit fill (alignNucleotidesEnd2EndSseU8's fixed point, aligner_swsse_ee_u8.cpp:775-1146), band form.
//
// Only the diagonals a valid alignment can touch are computed (EeBand, bt2g_align.hpp).  Lane l owns the 2 * RP consecutive
// diagonals dd = l * 2RP ..., the wave walks down the rows: in row i the lane's cells are the columns j = i + dd - lo.
//  * the diagonal predecessor of a cell is the lane's own value of the previous row (no shuffle);
//  * the cell above is the next diagonal of the previous row: a register shift inside the lane, one DPP wave shift across;
//  * the cell to the left is the previous diagonal of the SAME row.  E(dd) = max(E(dd-1) - rdgape, H(dd-1) - rdgapo) is a
//    max-plus prefix over the row; because rdgapo >= rdgape, opening from an H that itself came from E never beats extending
//    that E, so E is a scan over Hd = max(diagonal, F) alone: lane-local chain, then a 6-step DPP scan of the lane carries
//    (row_shr 1/2/4/8, row_bcast 15/31) with the per-lane decay 2RP * rdgape, then one more local pass.
// Scores are unsigned 16-bit values, two diagonals per register (v_pk_sub_u16 clamp = the reference's saturating subtract).
// The substitution penalty is one v_perm_b32: the lane keeps its reference characters as byte selectors (code | 0x0c00 per
// half) into the row's 5-entry penalty table, which is wave-uniform (A,C,G,T in one scalar register, N in another).
//
// PRED = false: score-only pass, nothing is stored -- just the best last-row score.  Most DP problems of a repeat-rich read
// fail (best < minsc: up to -D of them in a row) and a failed problem is never backtraced.  The pass stops at the first row
// (checked every 4th) in which no cell still holds minsc: end to end there is no match bonus, so scores only fall.
// PRED = true: the full fill.  What goes to memory is ONE BYTE per cell saying which predecessors are score-consistent (PB_*
// in bt2g_align.hpp) -- exactly the questions the reference's backtrace asks of H/E/F (:1330-1520), answered while the
// neighbours are in registers.  Identities used to fold the five H options into two flags (gaps allowed in the row):
//   H == H_up - rfgapo  <=>  H == F and F == H_up - rfgapo      (F >= H_up - rfgapo and H >= F);  likewise for the
//   extension F_up - rfgape and for E with H_left / E_left.
// A cell left of column 0 has only zero inputs and stays zero, so column 0 needs no special case: with H_left = E_left =
// H_diag = 0 none of HD/HE/EO/EE can be set on a cell whose score is a real one (> 0).  Cells right of the last column hold
// junk that never flows back (every dependency points to a smaller or equal column).
// Row i of the matrix is bytes [i * 128RP, (i + 1) * 128RP): lane l stores its 2RP bytes at offset l * 2RP.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 p_make(unsigned short lo, unsigned short hi) { u16x2 v; v.x = lo; v.y = hi; return v; }     // (C++: `(u16x2)(a, b)` would be a comma expression)
__device__ __forceinline__ u16x2 p_splat(int v) { return p_make((unsigned short)v, (unsigned short)v); }
__device__ __forceinline__ u16x2 p_subs(u16x2 a, u16x2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ u16x2 p_max(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 p_min(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ unsigned short s_subs(unsigned short a, unsigned short b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ unsigned short s_max(unsigned short a, unsigned short b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t p_bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 p_from(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ u16x2 p_ne(u16x2 a, u16x2 b) { return p_from(pk::nz(p_bits(a ^ b))); }     // 1 where different, else 0 (pk::nz: v_pk_min_u16 spelled out, see there)
// (lo.y, hi.x): the pair one diagonal further along
__device__ __forceinline__ uint32_t shift_in(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
// DPP moves: lanes without a source get 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true); }
constexpr int kDppRowShr = 0x110, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138, kDppBcast15 = 0x142, kDppBcast31 = 0x143;

// W16: the same recurrence in the reference's 16-bit representation (alignNucleotidesEnd2EndSseI16, aligner_swsse_ee_i16.cpp:780-1146: scores biased
// by 0x7fff in signed 16 bits, -32768 = minus infinity, saturating subtraction, barrier rows veto gap opens / extensions) -- as unsigned 16-bit
// values that is a bias of 0xffff instead of 0xff and nothing else: the same saturating arithmetic, the same seven questions per cell.  (Round 6:
// reads whose minimum score is below -254, i.e. longer than 423 bp at the default threshold, used to be filled on the anti-diagonal wavefront
// with 8 bytes per cell of the whole rectangle, fill_ee_i16_wave below; that form stays for bands wider than 2 048 diagonals.)
template <int RP, bool PRED, bool W16>
__device__ __forceinline__ int fill_ee_u8_band(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols, int lo, int thr, uint8_t* __restrict__ pm, uint32_t& rows_done) {
	constexpr uint32_t kBias = W16 ? 0xffffu : 0xffu;
	constexpr int kLastLo = W16 ? -32768 : -0xff;      // last-row score of a column the band does not reach (the scores go to a 16-bit row: < any minimum score)
	rows_done = rows;
	constexpr int N2 = 2 * RP;
	constexpr uint32_t W = 128u * RP;
	const int lane = threadIdx.x & 63;
	const int dd0 = lane * N2;
	const int j00 = dd0 - lo;                 // column of the lane's first diagonal in row 0
	uint32_t refS[RP];                        // reference characters of the lane's cells as v_perm selectors
	u16x2 Hp[RP], Fp[RP];                     // H and F of the previous row
#pragma unroll
	for (int k = 0; k < RP; k++) {
		uint32_t sel = 0, h = 0;
#pragma unroll
		for (int hh = 0; hh < 2; hh++) {
			const int j = j00 + 2 * k + hh;
			const bool real = (uint32_t)j < cols;
			const uint32_t code = real ? (uint32_t)__builtin_ctz((uint32_t)dev_rf()[real ? j : 0] | 16u) : 4u;
			sel |= (code | 0x0c00u) << (16 * hh);
			h |= (real ? kBias : 0u) << (16 * hh);       // "row -1": an alignment may start in any column of row 0
		}
		refS[k] = sel; Hp[k] = p_from(h); Fp[k] = p_splat(0);
	}
	if (PRED) for (uint32_t j = (uint32_t)lane; j < cols; j += 64) dev_lastrow()[j] = (int16_t)kLastLo;      // columns the band does not reach in the last row
	// (the parameter block is read from LDS: its fields arrive in vector registers and count as lane-varying until said otherwise)
	const int rdgape = __builtin_amdgcn_readfirstlane(P.rdgape), rdgapo = __builtin_amdgcn_readfirstlane(P.rdgapo), gapbar = __builtin_amdgcn_readfirstlane(P.gapbar);
	const u16x2 rdoP = p_splat(rdgapo), rfoP = p_splat(__builtin_amdgcn_readfirstlane(P.rfgapo)), rfeP = p_splat(__builtin_amdgcn_readfirstlane(P.rfgape));
	uint32_t npen_word = (uint32_t)__builtin_amdgcn_readfirstlane(P.n_pen) & 0xffu;
	uint32_t k7f = 0x007f007fu;
	asm("" : "+v"(npen_word), "+v"(k7f));      // in vector registers: the row's v_perm_b32 / v_bitop3_b32 take another operand from a scalar one, and an instruction reads one scalar register only
	// The column that enters a lane's last diagonal in the next row is the column of the NEXT lane's first diagonal in this row: one DPP move.
	// Only the wave's last diagonal gets a column nobody holds yet, 64 * N2 - lo + i after row i: lane ii of every 64-row chunk fetches the one of its row.
	const int jin63 = 64 * N2 - lo;
	auto cap = [](uint32_t v) -> unsigned short { constexpr uint32_t m = W16 ? 0xffffu : 1023u; return (unsigned short)(v > m ? m : v); };     // scores are <= the bias: any larger decay is "to zero"
	// The scan over the lanes' carries: E decays by D = 2RP * rdgape from one lane to the next, so with Y_l = carry_l + l * D the decayed
	// maximum  max_{l' <= l} (carry_l' - (l - l') D)  is  (prefix-max of Y)_l - l * D  -- a PLAIN prefix maximum, whose DPP steps are
	// single v_max_u32 instructions with a DPP operand (no subtraction between the move and the max).  No saturation is needed: the term
	// l' = l alone keeps the maximum >= 0, and terms that would have saturated to 0 cannot win.
	const uint32_t D = (uint32_t)N2 * (uint32_t)rdgape;
	const uint32_t lD = (uint32_t)lane * D, lDprev = lane ? lD - D : 0u;      // (lane 0: the prefix of no lane, 0 - 0)
	u16x2 decP[RP];                           // decay of the carry-in on its way to each of the lane's diagonals
#pragma unroll
	for (int k = 0; k < RP; k++) decP[k] = p_make(cap((uint32_t)(2 * k) * (uint32_t)rdgape), cap((uint32_t)(2 * k + 1) * (uint32_t)rdgape));
	u16x2 HP[RP];
#pragma unroll
	for (int k = 0; k < RP; k++) HP[k] = p_splat(0);
	for (uint32_t c0 = 0; c0 < rows; c0 += 64) {
		// the penalty tables of rows c0 .. c0+63, one per lane: byte b = penalty against reference character b
		uint32_t tab = 0;
		{
			const uint32_t ri = c0 + (uint32_t)lane;
			if (ri < rows) {
				const int c = rd_char(g_hot, g_hot.len, fw, ri);
				const int q = rd_qual(g_hot, g_hot.len, fw, ri) - 33;
				const uint32_t mm = (uint32_t)(c > 3 ? P.n_pen : mm_penalty(P, q < 0 ? 0 : q)) & 0xffu;
				tab = mm * 0x01010101u;
				if (c <= 3) tab &= ~(0xffu << (8 * c));
			}
		}
		uint32_t chq;
		{
			int jc = jin63 + (int)c0 + lane; jc = jc < 0 ? 0 : jc; if (jc > (int)cols) jc = (int)cols;      // outside the window: any character will do (see above)
			chq = (uint32_t)__builtin_ctz((uint32_t)dev_rf()[jc] | 16u) | 0x0c00u;
		}
		const uint32_t nrow = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rows - c0 < 64u ? rows - c0 : 64u));
		for (uint32_t ii = 0; ii < nrow; ii++) {
			const uint32_t i = c0 + ii;
			const uint32_t T1 = (uint32_t)__builtin_amdgcn_readlane((int)tab, (int)ii);
			const bool bar = (int)i < gapbar || (int)(rows - i - 1) < gapbar;       // no gaps this close to either end of the read
			const bool nof = bar || i == 0;      // F is vetoed: saturating subtraction of 0xffff, i.e. 0
			// ---- diagonal and vertical predecessors ----
			const uint32_t nxtH = dpp0<kDppWaveShl1, 0xf>(p_bits(Hp[0])), nxtF = dpp0<kDppWaveShl1, 0xf>(p_bits(Fp[0]));
			u16x2 penP[RP], HupP[RP], FupP[RP], fP[RP], HdP[RP];
#pragma unroll
			for (int k = 0; k < RP; k++) {
				penP[k] = p_from(__builtin_amdgcn_perm(npen_word, T1, refS[k]));
				HupP[k] = p_from(shift_in(p_bits(Hp[k]), k + 1 < RP ? p_bits(Hp[k + 1 < RP ? k + 1 : 0]) : nxtH));
				FupP[k] = p_from(shift_in(p_bits(Fp[k]), k + 1 < RP ? p_bits(Fp[k + 1 < RP ? k + 1 : 0]) : nxtF));
				fP[k] = nof ? p_splat(0) : p_max(p_subs(FupP[k], rfeP), p_subs(HupP[k], rfoP));
				HdP[k] = p_max(p_subs(Hp[k], penP[k]), fP[k]);
			}
			// ---- horizontal: E as a max-plus scan over the row ----
			u16x2 EP[RP];
			if (!bar) {
				unsigned short e0[N2];
				unsigned short carry = 0;
#pragma unroll
				for (int c = 0; c < N2; c++) {
					const u16x2 u = p_subs(HdP[c >> 1], rdoP);
					e0[c] = carry;
					carry = s_max(s_subs(carry, (unsigned short)rdgape), (c & 1) ? u.y : u.x);
				}
				uint32_t Y = (uint32_t)carry + lD;
				Y = umax32(Y, dpp0<kDppRowShr + 1, 0xf>(Y));
				Y = umax32(Y, dpp0<kDppRowShr + 2, 0xf>(Y));
				Y = umax32(Y, dpp0<kDppRowShr + 4, 0xf>(Y));
				Y = umax32(Y, dpp0<kDppRowShr + 8, 0xf>(Y));
				Y = umax32(Y, dpp0<kDppBcast15, 0xa>(Y));
				Y = umax32(Y, dpp0<kDppBcast31, 0xc>(Y));
				const uint32_t Pprev = dpp0<kDppWaveShr1, 0xf>(Y);      // prefix maximum up to the previous lane (0 into lane 0)
				const u16x2 einP = p_splat((int)(Pprev - lDprev));
#pragma unroll
				for (int k = 0; k < RP; k++) EP[k] = p_max(p_make(e0[2 * k], e0[2 * k + 1]), p_subs(einP, decP[k]));
			} else {
#pragma unroll
				for (int k = 0; k < RP; k++) EP[k] = p_splat(0);
			}
#pragma unroll
			for (int k = 0; k < RP; k++) HP[k] = p_max(HdP[k], EP[k]);
			if (PRED) {
				// left neighbours: the previous diagonal of this row
				const uint32_t prvH = dpp0<kDppWaveShr1, 0xf>(p_bits(HP[RP - 1])), prvE = dpp0<kDppWaveShr1, 0xf>(p_bits(EP[RP - 1]));
				const u16x2 rdeP = p_splat(rdgape);
				const uint32_t nogap = bar ? (uint32_t)(PB_HE | PB_HF) * 0x10001u : 0u;      // no gaps in this row: H is never "from E" / "from F"
				BT2_G uint8_t* rowp = (BT2_G uint8_t*)pm + (uint64_t)i * W + (uint32_t)dd0;
#pragma unroll
				for (int k = 0; k < RP; k++) {
					const u16x2 hlP = p_from(shift_in(k > 0 ? p_bits(HP[k > 0 ? k - 1 : 0]) : prvH, p_bits(HP[k])));
					const u16x2 elP = p_from(shift_in(k > 0 ? p_bits(EP[k > 0 ? k - 1 : 0]) : prvE, p_bits(EP[k])));
					const u16x2 h = HP[k], e = EP[k], f = fP[k];
					// one "differs" bit per question and half, the highest first ((n << 1) | bit: one instruction each, the halves do not meet -- 7 bits); inverted at the end
					uint32_t n = p_bits(p_ne(f + rfeP, FupP[k]));                                  // PB_FE
					n = pk::shl1_or(n, p_bits(p_ne(f + rfoP, HupP[k])));                           // PB_FO
					n = pk::shl1_or(n, p_bits(p_ne(e + rdeP, elP)));                               // PB_EE
					n = pk::shl1_or(n, p_bits(p_ne(e + rdoP, hlP)));                               // PB_EO
					n = pk::shl1_or(n, p_bits(p_ne(h, f)));                                        // PB_HF
					n = pk::shl1_or(n, p_bits(p_ne(h, e)));                                        // PB_HE
					n = pk::shl1_or(n, p_bits(p_ne(h + penP[k], Hp[k])));                          // PB_HD
					const uint32_t c = (n | nogap) ^ k7f;
					*reinterpret_cast<BT2_G uint16_t*>(rowp + 2 * k) = (uint16_t)__builtin_amdgcn_perm(0u, c, 0x0c0c0200u);      // bytes 0 and 2
				}
			}
			// ---- next row ----
#pragma unroll
			for (int k = 0; k < RP; k++) { Hp[k] = HP[k]; Fp[k] = fP[k]; }
			if (!PRED && (i & 3u) == 3u) {
				// scores only fall along a path (no match bonus end to end): once no cell of a row reaches minsc, no alignment will
				u16x2 m = HP[0];
#pragma unroll
				for (int k = 1; k < RP; k++) m = p_max(m, HP[k]);
				if (!__any((int)s_max(m.x, m.y) >= thr)) { rows_done = i + 1; return 0; }
			}
			{
				uint32_t newel = dpp0<kDppWaveShl1, 0xf>(refS[0]);
				{ const int sq = __builtin_amdgcn_readlane((int)chq, (int)ii); asm("v_writelane_b32 %0, %1, 63" : "+v"(newel) : "s"(sq)); }
#pragma unroll
				for (int k = 0; k < RP; k++) refS[k] = shift_in(refS[k], k + 1 < RP ? refS[k + 1 < RP ? k + 1 : 0] : newel);
			}
		}
	}
	// HP = the last row: best score over the window's columns; the scores themselves go to HOT.lastrow for the candidate gather
	int best = 0;
	const int jl0 = (int)rows - 1 + j00;
#pragma unroll
	for (int k = 0; k < RP; k++) {
#pragma unroll
		for (int hh = 0; hh < 2; hh++) {
			const int j = jl0 + 2 * k + hh;
			const int h = hh ? (int)HP[k].y : (int)HP[k].x;
			if ((uint32_t)j < cols) { best = imax(best, h); if (PRED) dev_lastrow()[j] = (int16_t)imax(h - (int)kBias, kLastLo); }
		}
	}
#pragma unroll
	for (int s = 32; s > 0; s >>= 1) best = imax(best, __shfl_xor(best, s));
	return best;
}

// The same recurrence in the reference's 16-bit representation (alignNucleotidesEnd2EndSseI16, aligner_swsse_ee_i16.cpp:780-1146):
// scores biased by 0x7fff, -32768 = minus infinity, saturating subtraction, barrier rows veto gap opens/extensions.
constexpr int kLo = -32768;
__device__ __forceinline__ int subs16(int a, int b) { const int v = a - b; return v < kLo ? kLo : v; }

template <int R>
__device__ __forceinline__ int fill_ee_i16_wave(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols,
                                               uint64_t* __restrict__ scratch) {
	const int lane = threadIdx.x & 63;
	const uint32_t nlanes = (rows + R - 1) / R;
	int rdc[R], mmp[R], veto[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		const uint32_t i = (uint32_t)lane * R + r;
		const bool valid = i < rows;
		rdc[r] = valid ? rd_char(g_hot, g_hot.len, fw, i) : 4;
		const int q = valid ? rd_qual(g_hot, g_hot.len, fw, i) - 33 : 0;
		mmp[r] = mm_penalty(P, q < 0 ? 0 : q);
		veto[r] = (valid && ((int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar)) ? 1 : 0;
	}
	int Hprev[R], Eprev[R];
#pragma unroll
	for (int r = 0; r < R; r++) { Hprev[r] = kLo; Eprev[r] = kLo; }
	int myHlast = kLo, myFlast = kLo, upHdiag = kLo, refm = 0, best = kLo;
	const uint32_t steps = cols + nlanes - 1;
	const bool lane_has_last = ((rows - 1) / R) == (uint32_t)lane;
	const int last_r = (int)((rows - 1) % R);
	for (uint32_t t = 0; t < steps; t++) {
		const int upH = __shfl_up(myHlast, 1);
		const int upF = __shfl_up(myFlast, 1);
		int upRef = __shfl_up(refm, 1);
		if (lane == 0) upRef = (t < cols) ? dev_rf()[t] : 16;
		refm = upRef;
		const int j = (int)t - lane;
		const bool active = j >= 0 && j < (int)cols && (uint32_t)lane < nlanes;
		int refc = 4;
		if (refm & 1) refc = 0; else if (refm & 2) refc = 1; else if (refm & 4) refc = 2; else if (refm & 8) refc = 3;
		int hdiag = (lane == 0) ? 0x7fff : (j == 0 ? kLo : upHdiag);
		int fin_h = upH, fin_f = upF;
		int Hnew[R], Enew[R], Fnew[R];
#pragma unroll
		for (int r = 0; r < R; r++) {
			int pen;
			if (rdc[r] > 3 || refc > 3) pen = P.n_pen; else pen = (rdc[r] == refc) ? -P.match_bonus : mmp[r];
			const int e = (j == 0) ? kLo : imax(subs16(Eprev[r], P.rdgape), veto[r] ? kLo : subs16(Hprev[r], P.rdgapo));
			int f;
			if (lane == 0 && r == 0) f = kLo;
			else f = veto[r] ? kLo : imax(subs16(fin_f, P.rfgape), subs16(fin_h, P.rfgapo));
			const int h = imax(imax(subs16(hdiag, pen), e), f);
			Hnew[r] = h; Enew[r] = e; Fnew[r] = f;
			hdiag = Hprev[r];
			fin_h = h; fin_f = f;
		}
		if (active) {
			uint64_t* base = scratch + ((uint64_t)t * R) * 64 + lane;     // 512 contiguous bytes per store
#pragma unroll
			for (int r = 0; r < R; r++) base[r * 64] = (uint64_t)(uint16_t)Hnew[r] | ((uint64_t)(uint16_t)Enew[r] << 16) | ((uint64_t)(uint16_t)Fnew[r] << 32);
#pragma unroll
			for (int r = 0; r < R; r++) { Hprev[r] = Hnew[r]; Eprev[r] = Enew[r]; }
			if (lane_has_last) best = imax(best, Hnew[last_r]);
		}
		upHdiag = upH;
		if (active) { myHlast = Hnew[R - 1]; myFlast = Fnew[R - 1]; }
	}
	return __shfl(best, (int)((rows - 1) / R));
}

// The same fill for reads of more than 512 rows (the long-read class: 16, 24 or 32 rows per lane, dp_R rounds up to these).  fill_ee_i16_wave's
// eight arrays of R registers do not fit at this R (and instantiating it for every R from 12 to 32 did not finish compiling, round 5): here a
// row's constants are one register (read character | mismatch penalty << 8 | gap veto << 16) and its previous H and E another (two 16-bit
// halves, as they are stored), the cells of a step are computed and stored one after the other.  Same values, same matrix layout (dp_cell).
#ifdef BT2G_KCLASS_LR
template <int R>
__device__ __attribute__((noinline)) int fill_ee_i16_leaf_big(bool fw_, uint32_t rows_, uint32_t cols_, uint64_t* m64_) {
	const bool fw = __builtin_amdgcn_readfirstlane((int)fw_) != 0;
	const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows_), cols = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols_);
	const uint64_t pa = (uint64_t)reinterpret_cast<uintptr_t>(m64_);
	uint64_t* scratch = reinterpret_cast<uint64_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa));
	const AlignParams& P = g_P;
	const int rdgapo = P.rdgapo, rdgape = P.rdgape, rfgapo = P.rfgapo, rfgape = P.rfgape, n_pen = P.n_pen, bonus = P.match_bonus;
	const int lane = threadIdx.x & 63;
	const uint32_t nlanes = (rows + R - 1) / R;
	uint32_t rowc[R], he[R];
#pragma unroll
	for (int r = 0; r < R; r++) {
		const uint32_t i = (uint32_t)lane * R + r;
		const bool valid = i < rows;
		const int rdc = valid ? rd_char(g_hot, g_hot.len, fw, i) : 4;
		const int q = valid ? rd_qual(g_hot, g_hot.len, fw, i) - 33 : 0;
		const int mmp = mm_penalty(P, q < 0 ? 0 : q);
		const int veto = (valid && ((int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar)) ? 1 : 0;
		rowc[r] = (uint32_t)rdc | ((uint32_t)(mmp & 0xff) << 8) | ((uint32_t)veto << 16);
		he[r] = 0x80008000u;       // H = E = minus infinity
	}
	int myHlast = kLo, myFlast = kLo, upHdiag = kLo, refm = 0, best = kLo;
	const uint32_t steps = cols + nlanes - 1;
	const bool lane_has_last = ((rows - 1) / R) == (uint32_t)lane;
	const int last_r = (int)((rows - 1) % R);
	for (uint32_t t = 0; t < steps; t++) {
		const int upH = __shfl_up(myHlast, 1);
		const int upF = __shfl_up(myFlast, 1);
		int upRef = __shfl_up(refm, 1);
		if (lane == 0) upRef = (t < cols) ? dev_rf()[t] : 16;
		refm = upRef;
		const int j = (int)t - lane;
		const bool active = j >= 0 && j < (int)cols && (uint32_t)lane < nlanes;
		int refc = 4;
		if (refm & 1) refc = 0; else if (refm & 2) refc = 1; else if (refm & 4) refc = 2; else if (refm & 8) refc = 3;
		int hdiag = (lane == 0) ? 0x7fff : (j == 0 ? kLo : upHdiag);
		int fin_h = upH, fin_f = upF;
		uint64_t* base = scratch + ((uint64_t)t * R) * 64 + lane;     // 512 contiguous bytes per store
		int hbest = kLo;
#pragma unroll
		for (int r = 0; r < R; r++) {
			const uint32_t rc = rowc[r];
			const int rdc = (int)(rc & 0xff), mmp = (int)((rc >> 8) & 0xff);
			const bool veto = (rc >> 16) != 0;
			const int hp = (int)(int16_t)(uint16_t)(he[r] & 0xffffu), ep = (int)(int16_t)(uint16_t)(he[r] >> 16);
			int pen;
			if (rdc > 3 || refc > 3) pen = n_pen; else pen = (rdc == refc) ? -bonus : mmp;
			const int e = (j == 0) ? kLo : imax(subs16(ep, rdgape), veto ? kLo : subs16(hp, rdgapo));
			int f;
			if (lane == 0 && r == 0) f = kLo;
			else f = veto ? kLo : imax(subs16(fin_f, rfgape), subs16(fin_h, rfgapo));
			const int h = imax(imax(subs16(hdiag, pen), e), f);
			hdiag = hp;
			fin_h = h; fin_f = f;
			const uint32_t lo = (uint32_t)(uint16_t)h | ((uint32_t)(uint16_t)e << 16);
			if (active) {
				base[r * 64] = (uint64_t)lo | ((uint64_t)(uint16_t)f << 32);
				he[r] = lo;
				if (r == last_r) hbest = h;
			}
		}
		if (active && lane_has_last) best = imax(best, hbest);
		upHdiag = upH;
		if (active) { myHlast = fin_h; myFlast = fin_f; }
	}
	return __shfl(best, (int)((rows - 1) / R));
}
#endif

// Local-mode fill (alignNucleotidesLocalSseU8 / ...I16; they agree wherever the 8-bit kernel does not saturate): plain scores, floor 0,
// two cells per register (bt2g_local_pk.hpp: the cell arithmetic and why a lane owns block `lane` in its low halves and block `lane + 64` in
// its high halves).  What leaves a cell is its predecessor byte (PB_*, with the local kernels' `> floor` rule: a neighbour whose score is 0 is no
// predecessor); bits nobody can look at are not cleaned up: the E / F bits of a cell whose E / F is 0 (the walk enters the E state of a cell
// only through an E above the floor, and there "neighbour - penalty == E > 0" already says the neighbour is above the floor) and all bits of a
// cell whose H is 0 (no walk reaches it: the diagonal bit needs a diagonal above the floor, E / F steps need a positive E / F).
// The column maximum travels down the blocks with the column; the last block sees every column complete, in order, and the per-column
// bookkeeping of the kernels (best score, lastsolcol, bail-out point, "the 8-bit kernel would have saturated") is scalar code on its value.
// EMIT: the candidate cells of the gather (gatherCellsNucleotidesLocalSseU8, aligner_swsse_loc_u8.cpp:1389-1496: score >= minsc, at or
// below the first row that can reach minsc, a match whose diagonal successor is not) are recognised while the cell is in registers and
// appended to `emit` (unsorted, columns beyond lastsolcol included: the gather drops those) -- the matrix is not read again to find them.
// PRED = false: the score-only pass (round 6).  Nine of ten local DP windows of a repeat-rich read never reach the minimum score (11.4 of 13.0 per read
// on the 400-bp workload) and are never backtraced: the seven predecessor flags -- 21 of the 35 instructions of a step -- and the byte stores are
// wasted on them.  The pass computes the same scores and the same bail-out column, nothing is stored; only a window whose best score reaches the
// minimum is filled again with PRED = true.  Both passes stop at the bail-out column (what lies behind it nobody looks at: lastsolcol_).
template <int RB, bool EMIT, bool PRED = true>
__device__ __forceinline__ int fill_local_pk(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols, uint8_t* __restrict__ pm,
                                             int minsc, uint32_t& lastsolcol, uint32_t& sat8, BT2_G BtCand* emit, uint32_t emit_cap, uint32_t& n_emit) {
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t nblocks = (rows + RB - 1) / RB;      // <= 128
	LocalPkRows<RB> K;
	uint32_t rdn[RB], cok[RB];      // EMIT: 1 << character of the row below (0 under the last row); 1 where the row can hold a candidate
	int bias = P.n_pen;
	const uint32_t minrow = EMIT ? (uint32_t)(((minsc + P.match_bonus - 1) / P.match_bonus) - 1) : 0u;
#pragma unroll
	for (int r = 0; r < RB; r++) {
		uint32_t km = 0, kb = 0, kn = 0, kv = 0, ko = 0, kd = 0, kc = 0;
#pragma unroll
		for (int half = 0; half < 2; half++) {
			const uint32_t i = (lane + 64u * half) * RB + r;
			const bool valid = i < rows;
			const int rdc = valid ? rd_char(g_hot, g_hot.len, fw, i) : 4;
			const int q = valid ? rd_qual(g_hot, g_hot.len, fw, i) - 33 : 0;
			const int mmp = mm_penalty(P, q < 0 ? 0 : q);
			const bool veto = (int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar;
			if (valid && rdc <= 3 && mmp > bias) bias = mmp;
			const LocalPkRow1 c = local_pk_row(valid, rdc, mmp, veto, P.match_bonus, P.n_pen);
			const uint32_t below = (i + 1 < rows) ? (1u << rd_char(g_hot, g_hot.len, fw, i + 1)) : 0u;
			const uint32_t ok = (valid && i >= minrow) ? 1u : 0u;
			km |= c.rowmask << (16 * half); kb |= c.bpm << (16 * half); kn |= c.nmmp << (16 * half); kv |= c.vm << (16 * half); ko |= c.okm << (16 * half);
			kd |= below << (16 * half); kc |= ok << (16 * half);
		}
		K.rowmask[r] = km; K.bpm[r] = kb; K.nmmp[r] = kn; K.vm[r] = kv; K.okm[r] = ko; rdn[r] = kd; cok[r] = kc;
	}
	for (int o = 32; o > 0; o >>= 1) bias = imax(bias, __shfl_xor(bias, o));     // bias of the 8-bit query profile
	bias = __builtin_amdgcn_readfirstlane(bias);
	const LocalPkPen G = local_pk_pen(P.rdgapo, P.rdgape, P.rfgapo, P.rfgape, P.n_pen);
	// score >= minsc  <=>  (score + ge_add) -sat ge_sub != 0
	const uint32_t ge_add = pk::both(minsc <= 0 ? 1 : 0), ge_sub = pk::both(minsc <= 0 ? 0 : minsc - 1);
	uint32_t Hp[RB], Ep[RB];
#pragma unroll
	for (int r = 0; r < RB; r++) { Hp[r] = 0; Ep[r] = 0; }
	uint32_t myH = 0, myF = 0, mycm = 0, upHdiag = 0, refm = 0x00100010u;      // (a block that has not started sees reference N and zeros: it stays zero)
	int vmax = 0, lastsol = 0, sat = 0, bailed = 0;      // scalar: the last block's lane is the same for every step
	uint32_t nem = 0;
	const uint32_t last_blk = nblocks - 1, last_lane = last_blk & 63u, last_half = last_blk >> 6;
	const uint32_t steps = cols + nblocks - 1;
	for (uint32_t t = 0; t < steps; t++) {
		// what the blocks above computed in the previous step: lane - 1 for both halves, except block 64 (lane 0, high) which follows block 63
		// (lane 63, low); block 0 has nothing above it and is fed the reference
		uint32_t upH = (uint32_t)__shfl_up((int)myH, 1), upF = (uint32_t)__shfl_up((int)myF, 1), upM = (uint32_t)__shfl_up((int)mycm, 1), upR = (uint32_t)__shfl_up((int)refm, 1);
		{
			const uint32_t wH = (uint32_t)__builtin_amdgcn_readlane((int)myH, 63), wF = (uint32_t)__builtin_amdgcn_readlane((int)myF, 63);
			const uint32_t wM = (uint32_t)__builtin_amdgcn_readlane((int)mycm, 63), wR = (uint32_t)__builtin_amdgcn_readlane((int)refm, 63);
			const uint32_t feed = (t < cols) ? (uint32_t)dev_rf()[t] : 16u;
			if (lane == 0) { upH = wH << 16; upF = wF << 16; upM = wM << 16; upR = (wR << 16) | feed; }
		}
		refm = upR;
		uint32_t pb[RB];
		uint32_t cm = upM;
		local_pk_step<RB>(K, G, refm, upHdiag, upH, upF, Hp, Ep, pb, myH, myF, cm);
		mycm = cm;
		upHdiag = upH;
		const uint32_t j_lo = t - lane, j_hi = t - lane - 64u;      // (unsigned: negative = huge)
		const bool act_lo = j_lo < cols && lane < nblocks, act_hi = j_hi < cols && lane + 64u < nblocks;
		if (PRED) {
			uint8_t* base = pm + ((uint64_t)t * RB) * 128 + lane;
			if (act_lo) {
#pragma unroll
				for (int r = 0; r < RB; r++) base[r * 128] = (uint8_t)pb[r];
			}
			if (act_hi) {
#pragma unroll
				for (int r = 0; r < RB; r++) base[r * 128 + 64] = (uint8_t)(pb[r] >> 16);
			}
		}
		// end of column j = t - last_blk (aligner_swsse_loc_u8.cpp:1305-1335)
		if (t >= last_blk && !bailed) {
			const int j = (int)(t - last_blk);
			const uint32_t cmw = (uint32_t)__builtin_amdgcn_readlane((int)cm, last_lane);
			const int c = (int)(last_half ? cmw >> 16 : cmw & 0xffffu);
			if (c > vmax) vmax = c;
			if (c + bias >= 255) sat = 1;
			if (c < minsc) { if (c + (int)(cols - (uint32_t)j - 1) * P.match_bonus < minsc) bailed = 1; }
			else lastsol = j;
			if (!PRED && bailed) break;      // (the score-only pass has its answer; the storing pass goes on: the stage output and the CPU twin's check cover the whole rectangle)
		}
		if (EMIT) {
			// any cell of this step at or above minsc?  (rarely: most of the matrix is far below)
			uint32_t hm = Hp[0];
#pragma unroll
			for (int r = 1; r < RB; r++) hm = pk::maxu(hm, Hp[r]);
			const uint32_t actm = (act_lo ? 1u : 0u) | (act_hi ? 0x10000u : 0u);
			const uint32_t anyge = pk::nz(pk::subsu(pk::add(hm, ge_add), ge_sub)) & actm;
			if (__ballot(anyge != 0u)) {
				const uint32_t refn = (act_lo ? (uint32_t)dev_rf()[j_lo + 1] : 0u) | (act_hi ? (uint32_t)dev_rf()[j_hi + 1] << 16 : 0u);      // (column `cols` is padding)
#pragma unroll
				for (int r = 0; r < RB; r++) {
					const uint32_t h = Hp[r];
					const uint32_t cnd = pk::nz(pk::subsu(pk::add(h, ge_add), ge_sub)) & actm & cok[r] & pk::nz(refm & K.rowmask[r]) & (pk::nz(refn & rdn[r]) ^ 0x00010001u);
#pragma unroll
					for (int half = 0; half < 2; half++) {
						const bool c = ((cnd >> (16 * half)) & 1u) != 0u;
						const unsigned long long m = __ballot(c);
						if (m) {        // (wave-uniform)
							const uint32_t pos = nem + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
							if (c && pos < emit_cap) {
								BtCand v; v.score = (int32_t)((h >> (16 * half)) & 0xffffu); v.row = (uint16_t)((lane + 64u * half) * RB + r); v.col = (uint16_t)(half ? j_hi : j_lo);
								gst(emit + pos, v);
							}
							nem += (uint32_t)__popcll(m);
						}
					}
				}
			}
		}
	}
	lastsolcol = (uint32_t)lastsol;
	sat8 = (uint32_t)sat;
	n_emit = nem;
	return vmax;
}

// ---------------------------------------------------------------------------------------------------------------------
// The fills as LEAF functions, one real function per instantiation.  Why not one big dispatcher that inlines all of them: a function
// that is really called saves every callee-saved vector register it touches ANYWHERE in its body (v40-47, v56-63, ... 48 of the first
// 128) in its prologue and reloads them in its epilogue -- 256 bytes of scratch traffic per register and direction.  The dispatcher
// with all sixteen band fills inlined used all 128 registers (the RP = 16 instantiation needs them), so each of the ~8 DP windows of
// a read cost 2 x 25 KB of scratch traffic for nothing: round 3's PMC passes booked ~200 KB written per read to exactly this.  A leaf
// is allocated caller-saved registers first and the common instantiations (RP <= 2, R <= 3) fit into those 80: no prologue stores.
// Arguments of a real call arrive in vector registers and are lane-varying to the compiler: every leaf passes them through
// v_readfirstlane once; the parameter block is read from its LDS object by name.  The row the score-only pass stopped in comes back
// through g_st.fill_rows_done.
template <int RP, bool PRED, bool W16 = false>
__device__ __attribute__((noinline)) int fill_ee_u8_leaf(bool fw_, uint32_t rows_, uint32_t cols_, int lo_, int thr_, uint8_t* pm_) {
	const bool fw = __builtin_amdgcn_readfirstlane((int)fw_) != 0;
	const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows_), cols = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols_);
	const int lo = __builtin_amdgcn_readfirstlane(lo_), thr = __builtin_amdgcn_readfirstlane(thr_);
	const uint64_t pa = (uint64_t)reinterpret_cast<uintptr_t>(pm_);
	uint8_t* pm = reinterpret_cast<uint8_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa));
	uint32_t rows_done = rows;
	const int best = fill_ee_u8_band<RP, PRED, W16>(g_P, fw, rows, cols, lo, thr, pm, rows_done);
	if (!PRED && (threadIdx.x & 63) == 0) g_st.fill_rows_done = rows_done;
	return best;
}
template <int R>
__device__ __attribute__((noinline)) int fill_ee_i16_leaf(bool fw_, uint32_t rows_, uint32_t cols_, uint64_t* m64_) {
	const bool fw = __builtin_amdgcn_readfirstlane((int)fw_) != 0;
	const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows_), cols = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols_);
	const uint64_t pa = (uint64_t)reinterpret_cast<uintptr_t>(m64_);
	uint64_t* m64 = reinterpret_cast<uint64_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa));
	return fill_ee_i16_wave<R>(g_P, fw, rows, cols, m64);
}
// local fill: lastsolcol / sat8 / the number of emitted candidates come back through g_st (fill_lastsol, fill_sat8, n_emit)
template <int RB, bool EMIT, bool PRED = true>
__device__ __attribute__((noinline)) int fill_local_leaf(bool fw_, uint32_t rows_, uint32_t cols_, uint8_t* pm_, int ms_, BT2_G BtCand* emit_, uint32_t ecap_) {
	const bool fw = __builtin_amdgcn_readfirstlane((int)fw_) != 0;
	const uint32_t rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)rows_), cols = (uint32_t)__builtin_amdgcn_readfirstlane((int)cols_);
	const int ms = __builtin_amdgcn_readfirstlane(ms_);
	const uint32_t ecap = (uint32_t)__builtin_amdgcn_readfirstlane((int)ecap_);
	const uint64_t pa = (uint64_t)reinterpret_cast<uintptr_t>(pm_);
	uint8_t* pm = reinterpret_cast<uint8_t*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa));
	const uint64_t ea = (uint64_t)emit_;
	BT2_G BtCand* emit = (BT2_G BtCand*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ea >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ea));
	uint32_t lastsolcol = 0, sat8 = 0, nem = 0;
	const int best = fill_local_pk<RB, EMIT, PRED>(g_P, fw, rows, cols, pm, ms, lastsolcol, sat8, emit, ecap, nem);
	if ((threadIdx.x & 63) == 0) { g_st.fill_lastsol = lastsolcol; g_st.fill_sat8 = sat8; g_st.n_emit = nem; g_st.emit_vmax = best; }
	return best;
}

struct DevPlat {
	static __device__ __forceinline__ HotWork& hot() { return g_hot; }
	static __device__ __forceinline__ uint8_t* rf() { return dev_rf(); }
	static __device__ __forceinline__ Edit* ned() { return dev_ned(); }
	static __device__ __forceinline__ int16_t* lastrow() { return dev_lastrow(); }
	static __device__ __forceinline__ const AlignParams& params() { return g_P; }
	static __device__ __forceinline__ ReadParams& rparams() { return g_rp; }
	static __device__ __forceinline__ const PreComp* pre() { return &g_pre; }
	static __device__ __forceinline__ AlState& st() { return g_st; }
	// the wave's work area in HBM, typed as such (BT2_G, bt2g_device.hpp)
	static __device__ __forceinline__ BT2_G Work& work() { return *(BT2_G Work*)uni((uint64_t)g_st.wp); }      // (base in scalar registers: the addresses of WK's members are scalar + offset)
	template <typename TOff> static __device__ __forceinline__ const DevIndex<TOff>& index() { return *reinterpret_cast<const DevIndex<TOff>*>(g_ix_raw); }
	static __device__ __forceinline__ uint64_t clock() { return (uint64_t)wall_clock64(); }
	// The worker's control code computes the same value in every lane; uni() moves such a value into a
	// scalar register so that what is derived from it runs on the scalar ALU instead of 64 redundant lanes.
	// lane-strided loops of the worker: which lane this is, how many there are, "did any lane say yes"
	static __device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }
	static __device__ __forceinline__ uint32_t n_lanes() { return 64u; }
	static __device__ __forceinline__ bool any(bool b) { return __ballot(b) != 0ull; }
	static __device__ __forceinline__ void sync() { wave_fence(); }      // what the lanes wrote one entry each is read by all afterwards
	static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
	static __device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
	static __device__ __forceinline__ uint64_t uni(uint64_t v) {
		return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
	}
	static __device__ __forceinline__ int64_t uni(int64_t v) { return (int64_t)uni((uint64_t)v); }
	template <typename T> static __device__ __forceinline__ T* uni_ptr(T* p) { return reinterpret_cast<T*>(uni((uint64_t)reinterpret_cast<uintptr_t>(p))); }
	// SSEMatrix::initMasks for the whole rectangle: 16 bytes per lane per store (the region is 256-byte padded)
	static __device__ __forceinline__ void zero_masks(uint16_t* p, uint32_t n) {
		wave_fence();
		uint4* q = reinterpret_cast<uint4*>(p);
		const uint32_t n16 = (n + 7) / 8;
		const uint4 z = make_uint4(0, 0, 0, 0);
		for (uint32_t i = threadIdx.x & 63; i < n16; i += 64) q[i] = z;
		wave_fence();
	}
	static __device__ __forceinline__ void zero_u32(uint32_t* p, uint32_t n) {
		wave_fence();
		uint4* q = reinterpret_cast<uint4*>(p);
		const uint4 z = make_uint4(0, 0, 0, 0);
		for (uint32_t i = threadIdx.x & 63; i < (n + 3) / 4; i += 64) q[i] = z;
		wave_fence();
	}
	// ---- the register-only row sampler (Aligner::sample_rows_fast): lane j = range j, table entry e = lane e & 63 of register e >> 6 ----
	// RowSampler::init + Random1toN::init of up to 64 ranges at once: lane j reads range j -- one memory round trip for all ranges
	static __device__ __forceinline__ void samp_setup(const BT2_G SatPos* sat, uint32_t n, bool all_hits, uint32_t& mlo, uint32_t& mhi, uint32_t& rn, uint32_t& rthr, uint32_t& rfl,
	                                                  uint32_t& tlo, uint32_t& thi) {
		wave_fence();
		const uint32_t l = threadIdx.x & 63;
		double m = 0.0;
		rn = rthr = rfl = tlo = thi = 0;
		if (l < n) {
			const BT2_G SatPos* s = sat + l;
			const uint32_t size = gld(&s->size);
			const uint64_t topf = gld(&s->topf);
			m = samp_mass(gld(&s->nlex), gld(&s->nrex), size);
			uint32_t th = (uint32_t)(0.10f * (float)size); th = th > 16 ? th : 16;      // Random1toN::init (random_util.h:97-110)
			rn = size; rthr = th; rfl = (size < 128 || all_hits) ? 1u : 0u;
			tlo = (uint32_t)topf; thi = (uint32_t)(topf >> 32);
		}
		const uint64_t u = (uint64_t)__double_as_longlong(m);
		mlo = (uint32_t)u; mhi = (uint32_t)(u >> 32);
	}
	// running sums of the weights still in play, added left to right as RowSampler::next's scan adds them (so that "first range whose
	// running sum exceeds rd" is the same range); returns the total
	static __device__ __forceinline__ double prefix_live(uint32_t mlo, uint32_t mhi, uint64_t live, uint32_t& plo, uint32_t& phi) {
		const uint32_t l = threadIdx.x & 63;
		double acc = 0.0;
		uint64_t m = live;
		while (m) {
			const uint32_t i = (uint32_t)__builtin_ctzll(m);
			m &= m - 1;
			acc += f64_of((uint32_t)__builtin_amdgcn_readlane((int)mlo, (int)i), (uint32_t)__builtin_amdgcn_readlane((int)mhi, (int)i));
			if (l == i) { const uint64_t u = (uint64_t)__double_as_longlong(acc); plo = (uint32_t)u; phi = (uint32_t)(u >> 32); }
		}
		return acc;
	}
	static __device__ __forceinline__ uint32_t pick_prefix(uint32_t plo, uint32_t phi, uint64_t live, double rd) {
		const unsigned long long hit = __ballot(rd < f64_of(plo, phi)) & live;
		if (hit) return (uint32_t)__builtin_ctzll(hit);
		return live ? 63u - (uint32_t)__builtin_clzll(live) : 0xffffffffu;
	}
	// The table: K registers as ONE vector value per array (ext_vector_type: an SSA value, never memory -- a plain array of registers ends
	// up in scratch as soon as the optimiser merges two accesses into one with a run-time index, which it does).  Entry e sits in lane
	// e & 63 of element e >> 6; unused lanes hold key 0, which no entry has.
	template <int K> using LaneRegs = uint32_t __attribute__((ext_vector_type(K)));
	template <typename V> static __device__ __forceinline__ void tab_zero(V& t) { t = (V)(0u); }
	template <typename V> static __device__ __forceinline__ bool tab_lookup(const V& k, const V& v, uint32_t n, uint32_t key, uint32_t& val) {
		constexpr int K = sizeof(V) / 4;
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r * 64u >= n) break;
			const unsigned long long m = __ballot(k[r] == key);
			if (m) { val = (uint32_t)__builtin_amdgcn_readlane((int)v[r], (int)__builtin_ctzll(m)); return true; }
		}
		return false;
	}
	template <typename V> static __device__ __forceinline__ void tab_set(const V& k, V& v, uint32_t n, uint32_t key, uint32_t val) {
		constexpr int K = sizeof(V) / 4;
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r * 64u >= n) break;
			const unsigned long long m = __ballot(k[r] == key);
			if (m) { uint32_t t = v[r]; set_lane(t, (uint32_t)__builtin_ctzll(m), val); v[r] = t; return; }
		}
	}
	template <typename V> static __device__ __forceinline__ void tab_append(V& k, V& v, uint32_t& n, uint32_t key, uint32_t val) {
		constexpr int K = sizeof(V) / 4;
		const uint32_t hi = n >> 6, lo = n & 63u;
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r == hi) { uint32_t tk = k[r], tv = v[r]; set_lane(tk, lo, key); set_lane(tv, lo, val); k[r] = tk; v[r] = tv; }
		}
		n++;
	}
	// entries of kind | range `keyhi` whose payload is <= x
	template <typename V> static __device__ __forceinline__ uint32_t tab_count_le(const V& k, const V& v, uint32_t n, uint32_t keyhi, uint32_t x) {
		constexpr int K = sizeof(V) / 4;
		uint32_t c = 0;
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r * 64u >= n) break;
			c += (uint32_t)__popcll(__ballot((k[r] & 0xff000000u) == keyhi && v[r] <= x));
		}
		return c;
	}
	// the seen values of `range` (kind 2 entries) become kind 3 entries carrying value - rank: every lane counts, for each of its own
	// entries of the range, the range's values below it while those are broadcast one by one
	template <typename V> static __device__ __forceinline__ void tab_convert(V& k, V& v, uint32_t n, uint32_t range) {
		constexpr int K = sizeof(V) / 4;
		const uint32_t seenhi = (2u << 30) | (range << 24), convhi = (3u << 30) | (range << 24);
		V cnt = (V)(0u);
#pragma unroll
		for (int r2 = 0; r2 < K; r2++) {
			if ((uint32_t)r2 * 64u >= n) break;
			unsigned long long m = __ballot((k[r2] & 0xff000000u) == seenhi);
			while (m) {
				const uint32_t i = (uint32_t)__builtin_ctzll(m);
				m &= m - 1;
				const uint32_t sv = (uint32_t)__builtin_amdgcn_readlane((int)k[r2], (int)i) & 0xffffffu;
#pragma unroll
				for (int r = 0; r < K; r++) {
					if ((uint32_t)r * 64u >= n) break;
					cnt[r] += ((k[r] & 0xff000000u) == seenhi && (k[r] & 0xffffffu) > sv) ? 1u : 0u;
				}
			}
		}
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r * 64u >= n) break;
			if ((k[r] & 0xff000000u) == seenhi) { v[r] = (k[r] & 0xffffffu) - cnt[r]; k[r] = convhi | (k[r] & 0xffffffu); }
		}
	}
	// lane j <- the fields of range sat[j] the extension loop needs of a sampled row: rdoff | seedlen << 12 | fw << 18 | offidx << 20
	static __device__ __forceinline__ uint32_t range_fields(const BT2_G SatPos* sat, uint32_t n) {
		wave_fence();
		const uint32_t l = threadIdx.x & 63;
		uint32_t r = 0;
		if (l < uni(n)) { const BT2_G SatPos* s = sat + l; r = (gld(&s->rdoff) & 0xfffu) | ((gld(&s->seedlen) & 0x3fu) << 12) | ((uint32_t)(gld(&s->fw) != 0) << 18) | (gld(&s->offidx) << 20); }
		return r;
	}
	// ---- lane code (BT2_FOR_LANES / LV in bt2g_align_core.hpp): on the device the block runs once, a LaneReg is the lane's own register ----
	static __device__ __forceinline__ uint32_t lanes_first() { return threadIdx.x & 63; }
	static __device__ __forceinline__ uint32_t lanes_step() { return 64u; }
	static __device__ __forceinline__ uint32_t& lv(uint32_t& r, uint32_t) { return r; }
	static __device__ __forceinline__ const uint32_t& lv(const uint32_t& r, uint32_t) { return r; }
	static __device__ __forceinline__ uint64_t ballot(uint32_t r) { return (uint64_t)__ballot(r != 0u); }
	// sum of a lane register over the wave
	static __device__ __forceinline__ uint64_t lanes_sum(uint32_t r) {
		uint64_t v = r;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o);
		return uni(v);
	}
	static __device__ __forceinline__ uint32_t gather(uint32_t x, uint32_t idx) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)x); }
	template <typename V, typename F> static __device__ __forceinline__ void tab_for_each(const V& k, const V& v, uint32_t n_, F f, uint32_t from_ = 0u) {
		constexpr int K = sizeof(V) / 4;
		const uint32_t n = uni(n_), from = uni(from_);
#pragma unroll
		for (int r = 0; r < K; r++) {
			if ((uint32_t)r * 64u >= n) break;
			if ((uint32_t)r * 64u + 64u <= from) continue;
			const uint32_t cnt = n - (uint32_t)r * 64u < 64u ? n - (uint32_t)r * 64u : 64u;
			for (uint32_t i = from > (uint32_t)r * 64u ? from - (uint32_t)r * 64u : 0u; i < cnt; i++)
				f((uint32_t)r * 64u + i, (uint32_t)__builtin_amdgcn_readlane((int)k[r], (int)i), (uint32_t)__builtin_amdgcn_readlane((int)v[r], (int)i));
		}
	}
	template <typename V> static __device__ __forceinline__ void tab_set_at(V& v, uint32_t e_, uint32_t val) {
		constexpr int K = sizeof(V) / 4;
		const uint32_t e = uni(e_);
#pragma unroll
		for (int r = 0; r < K; r++) if ((uint32_t)r == (e >> 6)) { uint32_t t = v[r]; set_lane(t, e & 63u, val); v[r] = t; }
	}
	// the flagged lanes' (key, value) pairs become the next entries of the table, in lane order: destination slot n + j takes the pair of the
	// j-th flagged lane (found per destination lane by a select on the ballot mask, fetched with ds_bpermute)
	template <typename V> static __device__ __forceinline__ void tab_append_lanes(V& k, V& v, uint32_t& n_, uint32_t flag, uint32_t key, uint32_t val) {
		constexpr int K = sizeof(V) / 4;
		const uint64_t m = (uint64_t)__ballot(flag != 0u);
		if (!m) return;
		const uint32_t n = uni(n_), cnt = (uint32_t)__popcll(m), l = threadIdx.x & 63;
		const uint32_t j = (l - n) & 63u;           // this lane's slot is entry n + j of the table (when j < cnt)
		// select(m, j): the position of the j-th set bit
		uint64_t mm = m; uint32_t jj = j, pos = 0;
#pragma unroll
		for (uint32_t w = 32; w >= 1; w >>= 1) {
			const uint32_t c = (uint32_t)__popcll(mm & ((1ull << w) - 1ull));
			if (jj >= c) { jj -= c; mm >>= w; pos += w; }
		}
		const uint32_t sk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pos & 63u) << 2), (int)key), sv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pos & 63u) << 2), (int)val);
		const uint32_t slot = ((n + j) & ~63u) == (n & ~63u) ? (n & ~63u) + l : (n & ~63u) + 64u + l;      // the slot of this lane that lies in [n, n + 64)
		const bool take = j < cnt;
#pragma unroll
		for (int r = 0; r < K; r++) if (take && (slot >> 6) == (uint32_t)r) { k[r] = sk; v[r] = sv; }
		n_ = n + cnt;
	}
	// ---- the row sampler's hash tables in the launch's dynamic LDS (which holds per-window DP state otherwise: free while rows are being sampled) ----
	//   S:  the keys of the seen-list entries (a set: 4 bytes per slot)
	//   W:  swap-list entries, key -> index of the entry in the register table (8 bytes per slot)
	//   Bt: key -> the lanes of the batch in hand that write it (12 bytes per slot; cleared per batch)
	// Open addressing, linear probing, key 0 = empty; Bt at rf_at, W behind it, S behind that with whatever the launch has left.  sh_begin clears S and W
	// and says how many slots they have (nS | nW << 16; 0: the launch has no room -- no batching).
	typedef __attribute__((address_space(3))) uint32_t lds_u32;
	static constexpr uint32_t kShBatchSlots = 96u;
	static __device__ __forceinline__ lds_u32* lds_w(uint32_t a) { return (lds_u32*)(uintptr_t)a; }
	static __device__ __forceinline__ uint32_t sh_slot0(uint32_t key, uint32_t nslots) { return (((key * 2654435761u) >> 16) * nslots) >> 16; }
	static __device__ __forceinline__ void lds_zero(uint32_t at, uint32_t nwords) {
		wave_fence();
		for (uint32_t i = threadIdx.x & 63; i < nwords; i += 64) *lds_w(at + i * 4u) = 0u;
		wave_fence();
	}
	static __device__ __forceinline__ uint32_t sh_begin() {
		const uint32_t have = uni(g_st.dyn_bytes);
		if (have < kShBatchSlots * 12u + 1024u) return 0u;
		const uint32_t room = have - kShBatchSlots * 12u;
		// 8-byte slots for W, 4-byte slots for S: about 3 : 2 of the room (more than half of a typical read's draws are swap-list draws)
		uint32_t nW = (room * 3u / 5u) / 8u, nS = (room - nW * 8u) / 4u;
		if (nW > 1024u) nW = 1024u;
		if (nS > 4096u) nS = 4096u;
		lds_zero(uni(g_st.rf_at) + kShBatchSlots * 12u, nW * 2u + nS);
		return nS | (nW << 16);
	}
	static __device__ __forceinline__ void bh_clear(uint32_t) { lds_zero(uni(g_st.rf_at), kShBatchSlots * 3u); }
	// flagged lanes: the entry `key` (register-table index idx) enters S (seen-list keys, kind 2) or W (swap-list keys, kind 1); a key is put once
	static __device__ __forceinline__ void sh_put(uint32_t cfg, uint32_t key, uint32_t idx, uint32_t flag) {
		const uint32_t nS = cfg & 0xffffu, nW = cfg >> 16, w_at = uni(g_st.rf_at) + kShBatchSlots * 12u, s_at = w_at + nW * 8u;
		if (flag) {
			const bool seen = (key >> 30) == 2u;
			const uint32_t n = seen ? nS : nW, at = seen ? s_at : w_at, stride = seen ? 4u : 8u;
			uint32_t s = sh_slot0(key, n);
			for (uint32_t pr = 0; pr < n; pr++) {
				lds_u32* p = lds_w(at + s * stride);
				uint32_t expected = 0u;
				const bool ok = __hip_atomic_compare_exchange_strong(p, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (ok || expected == key) { if (!seen) p[1] = idx; break; }
				s = s + 1u == n ? 0u : s + 1u;
			}
		}
		wave_fence();
	}
	static __device__ __forceinline__ void sh_put1(uint32_t cfg, uint32_t key, uint32_t idx) { sh_put(cfg, uni(key), uni(idx), (threadIdx.x & 63) == 0 ? 1u : 0u); }
	// flagged lanes: found = key is in its table; idx = the entry's index in the register table (swap-list keys)
	static __device__ __forceinline__ void sh_get(uint32_t cfg, uint32_t key, uint32_t flag, uint32_t& found, uint32_t& idx) {
		const uint32_t nS = cfg & 0xffffu, nW = cfg >> 16, w_at = uni(g_st.rf_at) + kShBatchSlots * 12u, s_at = w_at + nW * 8u;
		found = 0u; idx = 0u;
		if (flag) {
			const bool seen = (key >> 30) == 2u;
			const uint32_t n = seen ? nS : nW, at = seen ? s_at : w_at, stride = seen ? 4u : 8u;
			uint32_t s = sh_slot0(key, n);
			for (uint32_t pr = 0; pr < n; pr++) {
				lds_u32* p = lds_w(at + s * stride);
				const uint32_t k2 = p[0];
				if (k2 == key) { found = 1u; if (!seen) idx = p[1]; break; }
				if (k2 == 0u) break;
				s = s + 1u == n ? 0u : s + 1u;
			}
		}
	}
	// flagged lanes: this lane joins the set of lanes that write `key` in the batch in hand
	static __device__ __forceinline__ void bh_mark(uint32_t, uint32_t key, uint32_t flag) {
		const uint32_t at = uni(g_st.rf_at), l = threadIdx.x & 63;
		if (flag) {
			uint32_t s = sh_slot0(key, kShBatchSlots);
			for (uint32_t pr = 0; pr < kShBatchSlots; pr++) {
				lds_u32* p = lds_w(at + s * 12u);
				uint32_t expected = 0u;
				const bool ok = __hip_atomic_compare_exchange_strong(p, &expected, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				if (ok || expected == key) { __hip_atomic_fetch_or(p + 1u + (l >> 5), 1u << (l & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
				s = s + 1u == kShBatchSlots ? 0u : s + 1u;
			}
		}
		wave_fence();
	}
	// flagged lanes: the lanes that write `key` in this batch (bit per lane)
	static __device__ __forceinline__ void bh_get(uint32_t, uint32_t key, uint32_t flag, uint32_t& lo, uint32_t& hi) {
		const uint32_t at = uni(g_st.rf_at);
		lo = hi = 0u;
		if (flag) {
			uint32_t s = sh_slot0(key, kShBatchSlots);
			for (uint32_t pr = 0; pr < kShBatchSlots; pr++) {
				lds_u32* p = lds_w(at + s * 12u);
				const uint32_t k2 = p[0];
				if (k2 == key) { lo = p[1]; hi = p[2]; break; }
				if (k2 == 0u) break;
				s = s + 1u == kShBatchSlots ? 0u : s + 1u;
			}
		}
	}
	// per lane: the value word of table entry idx (idx < 64 K)
	template <typename V> static __device__ __forceinline__ uint32_t tab_gather(const V& v, uint32_t idx) {
		constexpr int K = sizeof(V) / 4;
		uint32_t r = 0;
#pragma unroll
		for (int q = 0; q < K; q++) {
			if (__ballot((idx >> 6) == (uint32_t)q) == 0ull) continue;
			const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)v[q]);
			if ((idx >> 6) == (uint32_t)q) r = g;
		}
		return r;
	}
	// rows drawn by the sampler -> Work::srows, one 16-byte record per lane
	static __device__ __forceinline__ void flush_samp_rows(BT2_G SampRow* dst, uint32_t lo, uint32_t hi, uint32_t src, uint32_t cnt) {
		const uint32_t l = threadIdx.x & 63;
		if (l < cnt) { SampRow r; r.topf = ((uint64_t)hi << 32) | lo; r.src = src; r.done = 0; gst(dst + l, r); }
	}
	// Ebwt::getOffset for up to 64 rows at once, one LF walk per lane (joff_pack: offset + steps taken)
	template <typename TOff>
	static __device__ __forceinline__ void resolve_rows(const DevEbwt<TOff>& e, const SampRow* rows, uint32_t n, uint64_t* out) {
		wave_fence();
		const uint32_t lane = threadIdx.x & 63;
		if (lane < n) {
			uint32_t steps = 0;
			const TOff joff = bt2g::get_offset(e, (TOff)rows[lane].topf, steps);
			out[lane] = joff_pack((uint64_t)joff, steps);
		}
		wave_fence();
	}
	// is v among p[0..n)?  all lanes look at once (seen list of Random1toN)
	static __device__ __forceinline__ bool contains_u32(const uint32_t* p, uint32_t n, uint32_t v) {
		wave_fence();
		const uint32_t lane = threadIdx.x & 63;
		for (uint32_t base = 0; base < n; base += 64) {
			const uint32_t i = base + lane;
			if (__ballot(i < n && gld(p + (i < n ? i : 0)) == v)) return true;
		}
		return false;
	}
	// out[] <- the values of [0, n) that are not in seen[0..nseen), ascending (the swap list a Random1toN converts to).
	// 2048 values per pass: lane l keeps bitmap word l of the pass (built from the seen list, which every lane walks through
	// register broadcasts), then 64 values at a time are tested and compacted with ballot / popcount.
	static __device__ __attribute__((noinline)) void unseen_list(const uint32_t* seen_, uint32_t nseen_, uint32_t n_, uint32_t* out_) {
		wave_fence();
		const uint32_t* seen = uni_ptr(seen_); uint32_t* out = uni_ptr(out_);
		const uint32_t nseen = uni(nseen_), n = uni(n_);
		const uint32_t lane = threadIdx.x & 63;
		uint32_t count = 0;
		for (uint32_t p0 = 0; p0 < n; p0 += 2048) {
			uint32_t bm = 0;
			for (uint32_t base = 0; base < nseen; base += 64) {
				const uint32_t mine = base + lane < nseen ? gld(seen + base + lane) : 0xffffffffu;
				const uint32_t cnt = nseen - base < 64u ? nseen - base : 64u;
				for (uint32_t t = 0; t < cnt; t++) {
					const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)t) - p0;
					if (v < 2048u && (v >> 5) == lane) bm |= 1u << (v & 31);
				}
			}
			const uint32_t pend = n - p0 < 2048u ? n - p0 : 2048u;
			for (uint32_t b = 0; b < pend; b += 64) {
				const uint32_t j = b + lane;
				const uint32_t word = (uint32_t)__shfl((int)bm, (int)(j >> 5));
				const bool unseen = j < pend && !((word >> (j & 31)) & 1u);
				const unsigned long long m = __ballot(unseen);
				if (unseen) gst(out + count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), p0 + j);
				count += (uint32_t)__popcll(m);
			}
		}
		wave_fence();
	}
	static __device__ __forceinline__ void iota_u32(uint32_t* p, uint32_t n) {
		wave_fence();
		for (uint32_t i = threadIdx.x & 63; i < n; i += 64) gst(p + i, i);
		wave_fence();
	}
	static __device__ __forceinline__ void copy_words(void* dst, const void* src, uint32_t nwords) {
		wave_fence();
		const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
		uint32_t* d = reinterpret_cast<uint32_t*>(dst);
		for (uint32_t i = threadIdx.x & 63; i < nwords; i += 64) gst(d + i, gld(s + i));      // (both records live in the arena)
		wave_fence();
	}
	static __device__ __forceinline__ void set_epoch(uint32_t* p, uint32_t e) { if ((threadIdx.x & 63) == 0) *p = e; wave_fence(); }
	// Tile of the pred format anchored at (row, col): lane d <- predecessor byte and (epoch-checked) mask of the d-th cell from there in
	// direction `dir`: 0 = up the diagonal (row-d, col-d), 1 = left along the row (row, col-d), 2 = up the column (row-d, col).
	// One gather per plane: a single memory latency for up to 64 steps of a diagonal run -- or of a gap (the candidates next to an
	// alignment's end column all walk a gap of growing length back to its path; a tile per gap, not per gap position).
	// mk = the cell's reportedThrough bit.
	static __device__ __forceinline__ void bt_tile_pred(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t row, uint32_t col, uint32_t epoch, uint32_t dir,
	                                                    uint32_t& pr, uint32_t& mk) {
		wave_fence();        // marks of earlier steps -> visible to whichever lane re-reads them
		const uint32_t d = threadIdx.x & 63;
		uint32_t p = 0, m = 0;
		const uint32_t dr = dir == 1 ? 0u : d, dc = dir == 2 ? 0u : d;
		const uint32_t dd = (uint32_t)((int32_t)col - (int32_t)row + band_lo) + dr - dc;      // diagonal of the lane's cell (wraps past the band's edge)
		const bool ok = dr <= row && dc <= col && (band_w == 0u || dd < band_w);
		if (uni(g_st.rt_cur)) {
			// marks on chip (band form only: band_w > 0)
			if (ok) {
				const uint32_t r = row - dr, bit = r * band_w + dd;
				p = gld(reinterpret_cast<const uint8_t*>(dp.mat) + (uint64_t)r * band_w + dd);
				m = (dev_rt()[bit >> 5] >> (bit & 31u)) & 1u;
			}
		} else if (ok) {
			const uint64_t idx = pred_at(band_lo, band_w, row - dr, col - dc);
			p = gld(reinterpret_cast<const uint8_t*>(dp.mat) + idx);
			const uint32_t w = gld(dp.pmask + idx);
			m = (w >> kEpochShift) == epoch ? (w & 1u) : 0u;
		}
		pr = p; mk = m;
	}
	// ---- the candidates that die within a few cells, side by side (Aligner::next_alignment_m) ----
	// Band matrices (w > 0).  Marks on chip (one bit per cell in LDS, rt_cur) or epoch-tagged words in the arena.
	static __device__ __forceinline__ bool marks_batchable() { return true; }
	// per lane: the predecessor bytes (plo: cells 0-3, phi: cells 4-7) and reportedThrough bits (mk, bit i) of the eight cells from the lane's own cell on in
	// its own direction (0 up the diagonal, 1 left along the row, 2 up the column); a cell outside the band reads as 0 / unmarked.  Eight independent
	// loads per plane: one memory latency for eight steps of every candidate's path.
	static __device__ __forceinline__ void pred_tile8(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t row, uint32_t col, uint32_t dir, uint32_t flag,
	                                                  uint32_t& plo, uint32_t& phi, uint32_t& mk) {
		wave_fence();
		const uint8_t* pm = reinterpret_cast<const uint8_t*>(dp.mat);
		const bool chip = uni(g_st.rt_cur) != 0u;
		uint32_t pb[8], mw[8];
#pragma unroll
		for (uint32_t i = 0; i < 8; i++) {
			const uint32_t r = row - (dir == 1u ? 0u : i), c = col - (dir == 2u ? 0u : i);
			const uint32_t dd = (uint32_t)((int32_t)c - (int32_t)r + band_lo);
			const bool ok = flag && (int32_t)r >= 0 && (int32_t)c >= 0 && dd < band_w;
			const uint32_t bit = r * band_w + dd;
			pb[i] = ok ? (uint32_t)gld(pm + (uint64_t)r * band_w + dd) : 0u;
			if (chip) mw[i] = ok ? (dev_rt()[bit >> 5] >> (bit & 31u)) & 1u : 0u;
			else { const uint32_t w = ok ? gld(dp.pmask + (uint64_t)r * band_w + dd) : 0u; mw[i] = (w >> kEpochShift) == epoch ? (w & 1u) : 0u; }
		}
		plo = pb[0] | (pb[1] << 8) | (pb[2] << 16) | (pb[3] << 24);
		phi = pb[4] | (pb[5] << 8) | (pb[6] << 16) | (pb[7] << 24);
		mk = mw[0] | (mw[1] << 1) | (mw[2] << 2) | (mw[3] << 3) | (mw[4] << 4) | (mw[5] << 5) | (mw[6] << 6) | (mw[7] << 7);
	}
	// per lane: reportedThrough of the lane's own cell / set it
	static __device__ __forceinline__ uint32_t marks_of_cells(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t r, uint32_t c, uint32_t flag) {
		wave_fence();
		const uint32_t dd = (uint32_t)((int32_t)c - (int32_t)r + band_lo);
		const bool ok = flag && (int32_t)r >= 0 && (int32_t)c >= 0 && dd < band_w;
		const uint32_t bit = r * band_w + dd;
		if (uni(g_st.rt_cur)) return ok ? (dev_rt()[bit >> 5] >> (bit & 31u)) & 1u : 0u;
		const uint32_t w = ok ? gld(dp.pmask + (uint64_t)r * band_w + dd) : 0u;
		return (w >> kEpochShift) == epoch ? (w & 1u) : 0u;
	}
	static __device__ __forceinline__ void mark_cells(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t r, uint32_t c, uint32_t flag) {
		const uint32_t dd = (uint32_t)((int32_t)c - (int32_t)r + band_lo);
		const bool ok = flag && (int32_t)r >= 0 && (int32_t)c >= 0 && dd < band_w;
		const uint32_t bit = r * band_w + dd;
		if (uni(g_st.rt_cur)) { if (ok) __hip_atomic_fetch_or(dev_rt() + (bit >> 5), 1u << (bit & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
		else if (ok) gst(dp.pmask + (uint64_t)r * band_w + dd, 1u | (epoch << kEpochShift));
		wave_fence();
	}
	// setReportedThrough of one cell (wave-uniform)
	static __device__ __forceinline__ void rt_mark(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t row, uint32_t col) {
		if (uni(g_st.rt_cur)) {
			const uint32_t bit = row * band_w + (uint32_t)((int32_t)col - (int32_t)row + band_lo);
			if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(dev_rt() + (bit >> 5), 1u << (bit & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else gst(dp.pmask + pred_at(band_lo, band_w, row, col), 1u | (epoch << kEpochShift));
	}
	// ... of one cell per lane (the runs)
	static __device__ __forceinline__ void rt_mark_lane(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t r, uint32_t c) {
		if (uni(g_st.rt_cur)) {
			const uint32_t bit = r * band_w + (uint32_t)((int32_t)c - (int32_t)r + band_lo);
			__hip_atomic_fetch_or(dev_rt() + (bit >> 5), 1u << (bit & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else gst(dp.pmask + pred_at(band_lo, band_w, r, c), 1u | (epoch << kEpochShift));
	}
	// A band matrix with candidate cells is about to be walked (gather_cells, in place of SSEMatrix::initMasks): decide where its marks live.
	// On chip when the plane fits what the launch set aside (rows * w bits): cleared here.  Otherwise the epoch-tagged words in the arena.
	static __device__ __forceinline__ void rt_begin(const DpScratch& dp, uint32_t rows, bool band) {
		wave_fence();
		const uint32_t w = band ? uni(gld(dp.epoch + 2)) : 0u;
		const uint32_t rt_bytes = uni(g_st.rt_bytes);
		const bool on = w != 0u && rows * (w >> 3) <= rt_bytes;
		if (on) {
			uint4* q = reinterpret_cast<uint4*>(dev_rt());
			const uint32_t n16 = (rows * (w >> 3) + 15u) >> 4;
			for (uint32_t i = threadIdx.x & 63; i < n16; i += 64) q[i] = make_uint4(0, 0, 0, 0);
		}
		if ((threadIdx.x & 63) == 0) g_st.rt_cur = on ? 1u : 0u;
		wave_fence();
	}
	// seed hits of one pre-computed round (both strands) -> HOT.hits, one seed per lane
	static __device__ __forceinline__ void load_seed_hits(const bt2g_seed_hit* src_fw, const bt2g_seed_hit* src_rc, uint32_t nseeds, bool skip_fw, bool skip_rc) {
		for (int fwi = 0; fwi < 2; fwi++) {
			const bt2g_seed_hit* src = fwi == 0 ? src_fw : src_rc;
			const bool skip = fwi == 0 ? skip_fw : skip_rc;
			for (uint32_t i = threadIdx.x & 63; i < nseeds; i += 64) {
				HotHit h; h.topf = h.topb = 0; h.size = h.esize = 0;
				if (!skip) {
					const uint64_t topf = gld(&src[i].topf), botf = gld(&src[i].botf);
					if (botf > topf) { h.topf = topf; h.topb = gld(&src[i].topb); h.size = h.esize = (uint32_t)(botf - topf); }
				}
				g_hot.hits[fwi][i] = h;
				g_hot.sorted[fwi][i] = 0;
			}
		}
	}
	// The L <= 32 read characters of a seed as it aligns to the Watson strand, 2 bits each, first character in the top bits;
	// ok = no N among them.  One character per lane, OR-reduced.
	static __device__ __forceinline__ uint64_t seed_key(bool fw, uint32_t depth, uint32_t L, bool& ok) {
		const uint32_t k = threadIdx.x & 63;
		uint64_t part = 0;
		bool bad = false;
		if (k < L) {
			const int ch = fw ? (int)g_hot.seq[depth + k] : comp4(g_hot.seq[depth + L - 1 - k]);
			bad = ch > 3;
			part = (uint64_t)(ch & 3) << (2 * (L - 1 - k));
		}
		ok = __ballot(bad) == 0ull;
		uint32_t lo = (uint32_t)part, hi = (uint32_t)(part >> 32);
#pragma unroll
		for (int s = 32; s > 0; s >>= 1) { lo |= (uint32_t)__shfl_xor((int)lo, s); hi |= (uint32_t)__shfl_xor((int)hi, s); }
		return ((uint64_t)hi << 32) | lo;
	}
	// index of (key, len) among the n cached seed sequences, n if absent; 64 entries per round trip
	static __device__ __forceinline__ uint32_t find_key(const uint64_t* keys, const uint8_t* lens, uint32_t n, uint64_t key, uint8_t len) {
		const uint32_t lane = threadIdx.x & 63;
		for (uint32_t base = 0; base < n; base += 64) {
			const uint32_t i = base + lane;
			const bool hit = i < n && gld(keys + i) == key && gld(lens + i) == len;
			const unsigned long long m = __ballot(hit);
			if (m) return base + (uint32_t)__builtin_ctzll(m);
		}
		return n;
	}
	// Ebwt::joinedToTextOff (bt2_idx.cpp:113-171) for one wave-uniform offset: the fragment table is searched 64 ways at a time
	// (every lane probes one fragment start, a ballot picks the stride that holds the offset) instead of by a chain of
	// dependent binary-search reads -- 1 round for up to 64 fragments, 2 for up to 4096 -- then one gather for the record.
	template <typename TOff>
	static __device__ __forceinline__ void joined_to_text(const DevIndex<TOff>& ix, TOff qlen, TOff off, TOff& tidx, TOff& textoff, TOff& tlen,
	                                                      bool reject_straddle, bool& straddled) {
		straddled = false;
		tidx = (TOff)OffTraits<TOff>::kMask; textoff = 0; tlen = 0;
		const TOff n = ix.n_frag, len = ix.fw.len;
		if (off >= len || n == 0) return;
		const uint32_t lane = threadIdx.x & 63;
		const TOff* rs = ix.rstarts;
		TOff lo = 0, cnt = n;              // the answer is in [lo, lo + cnt) and rstarts[lo * 3] <= off
		while (cnt > 1) {
			const TOff stride = (cnt + 63) / 64;
			const TOff p = lo + (TOff)lane * stride;
			const bool le = p < lo + cnt && gld(rs + (uint64_t)p * 3) <= off;
			const unsigned long long m = __ballot(le) | 1ull;
			const uint32_t k = 63u - (uint32_t)__builtin_clzll(m);
			const TOff nlo = lo + (TOff)k * stride, end = lo + cnt;
			cnt = end - nlo < stride ? end - nlo : stride;
			lo = nlo;
		}
		const TOff elt = lo;
		// lanes 0-2: the record; lane 3: the start of the next fragment
		TOff v = 0;
		if (lane < 3) v = gld(rs + (uint64_t)elt * 3 + lane);
		else if (lane == 3) v = (elt == n - 1) ? len : gld(rs + ((uint64_t)elt + 1) * 3);
		const TOff lower = (TOff)__shfl((long long)v, 0), tid = (TOff)__shfl((long long)v, 1), fwoff = (TOff)__shfl((long long)v, 2), upper = (TOff)__shfl((long long)v, 3);
		if (off + qlen > upper) {
			straddled = true;
			if (reject_straddle) return;
		}
		tidx = tid;
		textoff = (off - lower) + fwoff;
		tlen = gld(ix.plen + tid);
		g_hot.frag_jlo = (uint64_t)lower; g_hot.frag_len = (uint64_t)(upper - lower); g_hot.frag_toff = (uint64_t)fwoff; g_hot.frag_tidx = (uint64_t)tid;
	}
	// is (ref, off, orient) inside one of the seen-diagonal intervals?  64 intervals per round trip
	static __device__ __forceinline__ bool diag_find(const DiagIval* d, uint32_t n, int32_t ref, int64_t off, int32_t orient) {
		const uint32_t lane = threadIdx.x & 63;
		for (uint32_t base = 0; base < n; base += 64) {
			const uint32_t i = base + lane;
			bool hit = false;
			if (i < n) {
				const int64_t o = gld(&d[i].off), l = gld(&d[i].len);
				const int32_t r = gld(&d[i].ref), t = gld(&d[i].orient);
				hit = r == ref && t == orient && off >= o && off < o + l;
			}
			if (__ballot(hit)) return true;
		}
		return false;
	}
	// Backtrace fast path (pred format, H state): the cells of the tile from lane td on that are "plain diagonal steps" --
	// not visited yet, HD the only consistent predecessor, not row 0 -- are walked in one go.  Every lane marks its own cell
	// (reportedThrough + the H choice, the word the step-by-step walk would leave: 3) and classifies its read/reference pair;
	// returns the run length, `info` = per lane readc << 4 | refm << 8 | qual << 16 | (N involved) << 1, `mm` = lanes whose
	// pair is not a match (they become edits, in lane order).
	static __device__ __forceinline__ uint32_t bt_diag_run(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t tile, uint32_t tile_hi,
	                                                       uint32_t td, uint32_t row, uint32_t col, bool fw, uint32_t rdlen, uint32_t maxl,
	                                                       uint32_t& info, uint64_t& mm) {
		const uint32_t d = threadIdx.x & 63;
		const uint32_t k = d - td;
		const bool in = d >= td && k < row && k <= col && k < maxl;
		const uint32_t pb = tile;
		const bool he = (pb & PB_HE) != 0, hf = (pb & PB_HF) != 0;
		const bool simple = in && tile_hi == 0 && (pb & PB_HD) && !(hf && (pb & (PB_FO | PB_FE))) && !(he && (pb & (PB_EO | PB_EE)));
		const unsigned long long sm = __ballot(simple) >> td;
		const uint32_t L = ~sm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~sm);
		info = 0; mm = 0;
		if (L == 0) return 0;
		bool edit = false;
		if (in && k < L) {
			const uint32_t r = row - k, c = col - k;
			const int readc = rd_char(g_hot, rdlen, fw, r);
			const int refm = dev_rf()[c];
			const int readq = rd_qual(g_hot, rdlen, fw, r);
			const int m = (refm >= 16 || readc > 3) ? -1 : (((1 << readc) & refm) ? 1 : 0);
			edit = m != 1;
			info = ((uint32_t)readc << 4) | ((uint32_t)refm << 8) | ((uint32_t)readq << 16) | (m == -1 ? 2u : 0u);
			rt_mark_lane(dp, band_lo, band_w, epoch, r, c);
		}
		mm = __ballot(edit);
		return L;
	}
	// Backtrace fast path inside a gap (pred format, E or F state, row / column tile): the cells from lane td on that can only extend the gap
	// -- not visited, no stored choice, E consistent with E-left alone (read gap) / F with F-up alone (reference gap) -- are walked in one go.
	// Every lane marks its cell (the word the step-by-step walk leaves: reportedThrough + "choice made, nothing else to try") and writes its
	// edit to ned[nned + k].  Returns the run length; core = some cell of the run lies on a core diagonal.
	static __device__ __forceinline__ uint32_t bt_gap_run(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t tile, uint32_t tile_hi,
	                                                      uint32_t td, uint32_t row, uint32_t col, bool read_gap, bool fw, uint32_t rdlen, uint32_t maxl,
	                                                      uint32_t nned, int r_triml, int r_corel, int r_corer, uint32_t& core) {
		const uint32_t d = threadIdx.x & 63;
		const uint32_t k = d - td;
		const bool in = d >= td && k < maxl && (read_gap ? k <= col : k < row);
		const uint32_t pb = tile;
		const uint32_t m = read_gap ? (pb >> 3) & 3u : (pb >> 5) & 3u;
		const bool ext = in && tile_hi == 0 && m == 2u;
		const unsigned long long sm = __ballot(ext) >> td;
		const uint32_t L = ~sm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~sm);
		core = 0;
		if (L == 0) return 0;
		bool incore = false;
		if (in && k < L) {
			const uint32_t r = read_gap ? row : row - k, c = read_gap ? col - k : col;
			Edit e;
			if (read_gap) { const int refm = dev_rf()[c]; e.pos = (uint16_t)(r + 1); e.chr = (uint8_t)((refm == 1 || refm == 2 || refm == 4 || refm == 8) ? code2chr(__builtin_ctz((unsigned)refm)) : 'N'); e.qchr = '-'; e.type = EDIT_READ_GAP; }
			else { e.pos = (uint16_t)r; e.chr = '-'; e.qchr = code2chr(rd_char(g_hot, rdlen, fw, r)); e.type = EDIT_REF_GAP; }
			e.pad = 0;
			dev_ned()[nned + k] = e;
			rt_mark_lane(dp, band_lo, band_w, epoch, r, c);
			const int diagi = (int)c - (int)r + r_triml;
			incore = diagi >= r_corel && diagi <= r_corer;
		}
		core = __ballot(incore) != 0ull ? 1u : 0u;
		return L;
	}
	// scores of the last DP row -> LDS (clamped at -32768; only scores >= minsc matter afterwards)
	static __device__ __forceinline__ void load_last_row(const uint32_t* mat, uint32_t R, uint32_t rows, uint32_t cols, bool wide) {
		wave_fence();
		for (uint32_t j = threadIdx.x & 63; j < cols; j += 64) {
			int sc;
			if (wide) sc = (int)(int16_t)(uint16_t)(reinterpret_cast<const uint64_t*>(mat)[dp_cell(R, rows - 1, j)] & 0xffff) - 0x7fff;
			else sc = (int)(mat[dp_cell(R, rows - 1, j)] & 0xff) - 0xff;
			dev_lastrow()[j] = (int16_t)(sc < -32768 ? -32768 : sc);
		}
		wave_fence();
	}
	// AlnRes copy, one 32-bit word per lane per pass; only the header and the edits in use move
	// stage a read (bases + qualities) into the hot state, lane-parallel
	static __device__ __forceinline__ void load_read(const uint8_t* seq, const uint8_t* qual, uint32_t len) {
		wave_fence();
		for (uint32_t i = threadIdx.x & 63; i < len; i += 64) { g_hot.seq[i] = seq[i]; g_hot.qual[i] = qual[i]; }
		wave_fence();
	}
	static __device__ __forceinline__ void copy_aln(BT2_G AlnRes& dst, const BT2_G AlnRes& src) {
		wave_fence();
		const uint32_t nb = (uint32_t)offsetof(AlnRes, ned) + (uint32_t)uni((uint32_t)src.nned) * (uint32_t)sizeof(Edit);
		const uint32_t nw = (nb + 3u) / 4u;
		const BT2_G uint32_t* s = (const BT2_G uint32_t*)&src;
		BT2_G uint32_t* d = (BT2_G uint32_t*)&dst;
		// (an odd number of 6-byte edits ends in the middle of a word: what lies behind it in the source is whatever an earlier alignment left there -- kept
		// out of the copy, so that the records of two runs are the same bytes)
		for (uint32_t i = threadIdx.x & 63; i < nw; i += 64) { uint32_t v = s[i]; if (i * 4u + 4u > nb) v &= 0xffffffffu >> (8u * (i * 4u + 4u - nb)); d[i] = v; }
		wave_fence();
	}
	// idx[0..n) <- the alignments' indices, descending by (score, index) -- AlnSinkWrap::selectByScore sorts (score, index) pairs ascending and reverses:
	// the scores go to `tmp`, every lane counts the alignments that come before each of its own
	static __device__ __attribute__((noinline)) void order_by_score(const BT2_G AlnRes* alns_, uint32_t n_, BT2_G uint32_t* idx_, BT2_G uint32_t* tmp_) {
		const BT2_G AlnRes* alns = uni_ptr(alns_); BT2_G uint32_t* idx = uni_ptr(idx_); BT2_G uint32_t* tmp = uni_ptr(tmp_);
		const uint32_t n = uni(n_), lane = threadIdx.x & 63;
		wave_fence();
		for (uint32_t i = lane; i < n; i += 64) gst(tmp + i, (uint32_t)gld(&alns[i].score));
		wave_fence();
		for (uint32_t base = 0; base < n; base += 64) {
			const uint32_t i = base + lane;
			const int32_t si = i < n ? (int32_t)gld(tmp + i) : 0;
			uint32_t rank = 0;
			for (uint32_t j = 0; j < n; j++) { const int32_t sj = (int32_t)gld(tmp + j); rank += (sj > si || (sj == si && j > i)) ? 1u : 0u; }
			if (i < n) gst(idx + rank, i);
		}
		wave_fence();
	}
	// Candidate cells of the last row, sorted by (score desc, col desc): every lane ranks its own cells
	// against the whole row (LDS broadcast reads) and writes them straight to their final slot.
	static __device__ __forceinline__ uint32_t gather_sort(BtCand* cands, uint32_t cap, uint32_t rows, uint32_t cols, int64_t minsc_dp) {
		const uint32_t lane = threadIdx.x & 63;
		const int thr = minsc_dp < -32768 ? -32768 : (int)minsc_dp;
		uint32_t total = 0;
		if (cols <= 256u) {
			// the common window (a seed extension: rows + 4 * maxhalf + 1 columns): every lane keeps the cells of its columns j = lane + 64 b as
			// keys (score | column: distinct, larger = earlier in the list; 0 = no candidate) and ranks them against the CANDIDATES only -- a few
			// dozen register broadcasts instead of a pass over every column of the row per batch of 64
			uint32_t key[4];
			unsigned long long cm[4];
#pragma unroll
			for (uint32_t b = 0; b < 4; b++) {
				const uint32_t j = b * 64u + lane;
				const int sc = j < cols ? (int)dev_lastrow()[j] : -65536;
				const bool is = j < cols && sc >= thr;
				key[b] = is ? (((uint32_t)(sc + 32768) << 16) | j) + 1u : 0u;
				cm[b] = __ballot(is);
				total += (uint32_t)__popcll(cm[b]);
			}
			if (total == 0) { wave_fence(); return 0; }
			uint32_t rank[4] = {0, 0, 0, 0};
#pragma unroll
			for (uint32_t b2 = 0; b2 < 4; b2++) {
				unsigned long long m = cm[b2];
				while (m) {
					const uint32_t i = (uint32_t)__builtin_ctzll(m);
					m &= m - 1;
					const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)key[b2], (int)i);
#pragma unroll
					for (uint32_t b = 0; b < 4; b++) rank[b] += k2 > key[b] ? 1u : 0u;
				}
			}
#pragma unroll
			for (uint32_t b = 0; b < 4; b++)
				if (key[b] != 0u && rank[b] < cap) { BtCand c; c.score = (int32_t)((key[b] - 1u) >> 16) - 32768; c.row = (uint16_t)(rows - 1); c.col = (uint16_t)((key[b] - 1u) & 0xffffu); cands[rank[b]] = c; }
			wave_fence();
			return total;
		}
		for (uint32_t base = 0; base < cols; base += 64) {
			const uint32_t j = base + lane;
			const int sc = j < cols ? (int)dev_lastrow()[j] : -65536;
			const bool is = j < cols && sc >= thr;
			if (__ballot(is) == 0ull) continue;
			uint32_t rank = 0;
			for (uint32_t k = 0; k < cols; k++) {
				const int s2 = (int)dev_lastrow()[k];
				if (s2 >= thr && (s2 > sc || (s2 == sc && k > j))) rank++;
			}
			if (is && rank < cap) { BtCand c; c.score = sc; c.row = (uint16_t)(rows - 1); c.col = (uint16_t)j; cands[rank] = c; }
			total += (uint32_t)__popcll(__ballot(is));
		}
		wave_fence();
		return total;
	}
	// Ebwt::getOffset for a wave-uniform row: one entry of the full suffix array (bt2g_device.hpp)
	template <typename TOff>
	static __device__ __forceinline__ TOff get_offset(const DevEbwt<TOff>& e, TOff row_, uint32_t& nsteps) {
		const uint64_t v = uni(gld(uni_ptr(e.sa) + uni((uint64_t)row_)));
		nsteps = (uint32_t)(v >> 48);
		return (TOff)(v & 0xffffffffffffull);
	}
	// A value per lane, kept in a vector register; lane(r, i) reads lane i's copy into a scalar register.
	using LaneReg = uint32_t;
	static __device__ __forceinline__ uint32_t lane(LaneReg r, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)r, (int)uni(i)); }
	static __device__ __forceinline__ void set_lane(LaneReg& r, uint32_t i, uint32_t v) {
		// (this clang has no __builtin_amdgcn_writelane; value and lane select are wave-uniform, i.e. scalar registers)
		// (gfx9 allows one scalar register per VALU instruction on the constant bus: the lane select goes through M0)
		asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(r) : "s"(uni(v)), "s"(uni(i)) : "m0");
	}
	static __device__ __forceinline__ void lanes_zero(LaneReg& r) { r = 0; }
	// lane i <- the 4 bytes at base[(word0 + i) * 4 ...] (0 past the end of the array); base is an LDS array
	static __device__ __forceinline__ LaneReg lanes_load(const uint8_t* base, uint32_t nbytes, uint32_t word0) {
		const uint32_t wd = word0 + (threadIdx.x & 63);
		return (wd * 4 + 4 <= nbytes) ? *reinterpret_cast<const uint32_t*>(base + wd * 4) : 0u;
	}
	// lane i <- candidate base + i (score; row | col << 16), 0 past the end of the list
	static __device__ __forceinline__ void lanes_load_cands(const BtCand* cands, uint32_t base, uint32_t n, LaneReg& w0, LaneReg& w1) {
		const uint32_t i = base + (threadIdx.x & 63);
		uint32_t a = 0, b = 0;
		if (i < n) { const uint32_t* p = reinterpret_cast<const uint32_t*>(cands + i); a = gld(p); b = gld(p + 1); }
		w0 = a; w1 = b;
	}
	// lane i <- p[base + i] (0 past n)
	static __device__ __forceinline__ LaneReg lanes_load_u32(const BT2_G uint32_t* p, uint32_t base, uint32_t n) {
		const uint32_t i = base + (threadIdx.x & 63);
		return i < n ? gld(p + i) : 0u;
	}
	// the first 64 entries of the seed cache's key table as lane registers (key halves, length | flags << 8, elements held)
	static __device__ __forceinline__ void lanes_load_keys(const BT2_G uint64_t* keys, const BT2_G uint8_t* lens, const BT2_G uint8_t* flags, const BT2_G uint32_t* eff, uint32_t n,
	                                                       LaneReg& klo, LaneReg& khi, LaneReg& klf, LaneReg& kef) {
		wave_fence();
		const uint32_t l = threadIdx.x & 63;
		klo = khi = klf = kef = 0;
		if (l < n) { const uint64_t k = gld(keys + l); klo = (uint32_t)k; khi = (uint32_t)(k >> 32); klf = (uint32_t)gld(lens + l) | ((uint32_t)gld(flags + l) << 8); kef = gld(eff + l); }
	}
	// index of (key, len) among the first n entries mirrored in the lanes, n if absent
	static __device__ __forceinline__ uint32_t find_key_lanes(LaneReg klo, LaneReg khi, LaneReg klf, uint32_t n, uint64_t key, uint8_t len) {
		const uint32_t l = threadIdx.x & 63;
		const unsigned long long m = __ballot(l < n && klo == (uint32_t)key && khi == (uint32_t)(key >> 32) && (klf & 0xffu) == (uint32_t)len);
		return m ? (uint32_t)__builtin_ctzll(m) : n;
	}
	// lane i holds candidate i of a batch as row | col << 16: flag the ones within sq rows and columns of the cell rc
	static __device__ __forceinline__ void dom_update(LaneReg& domv, LaneReg cw1, uint32_t rc, uint32_t sq) {
		const uint32_t orow = rc & 0xffffu, ocol = rc >> 16, row = cw1 & 0xffffu, col = cw1 >> 16;
		const uint32_t dr = row > orow ? row - orow : orow - row, dc = col > ocol ? col - ocol : ocol - col;
		domv |= (dr <= sq && dc <= sq) ? 1u : 0u;
	}
	// first candidate (lane) >= from of the nv in the batch that is not flagged, nv if none; low_first: one whose score is below minsc comes first
	static __device__ __forceinline__ uint32_t next_cand(LaneReg cw0, LaneReg domv, uint32_t from, uint32_t nv, int64_t minsc, bool& low_first) {
		const uint32_t l = threadIdx.x & 63;
		const bool in = l >= from && l < nv;
		const unsigned long long low = __ballot(in && (int64_t)(int32_t)cw0 < minsc), ok = __ballot(in && domv == 0u);
		const uint32_t fl = low ? (uint32_t)__builtin_ctzll(low) : 64u, fo = ok ? (uint32_t)__builtin_ctzll(ok) : 64u;
		low_first = fl < 64u && fl <= fo;
		return fo < nv ? fo : nv;
	}
	// is (row, col) within sq rows and sq columns of one of the first n cells (row | col << 16) held in the lanes of r?
	static __device__ __forceinline__ bool near_any(LaneReg r, uint32_t n, uint32_t row, uint32_t col, uint32_t sq) {
		const uint32_t l = threadIdx.x & 63;
		const uint32_t orow = r & 0xffffu, ocol = r >> 16;
		const uint32_t dr = row > orow ? row - orow : orow - row, dc = col > ocol ? col - ocol : ocol - col;
		return __ballot(l < n && dr <= sq && dc <= sq) != 0ull;
	}
	// Backtrace tile anchored at (row, col): lanes 0-15 cell(row-d, col-d), 16-31 cell(row-d-1, col-d),
	// 32-47 cell(row-d, col-d-1), 48-63 mask(row-d, col-d); one load instruction per array, one latency.
	static __device__ __forceinline__ void bt_tile(const DpScratch& dp, uint32_t R, uint32_t cols, uint32_t row, uint32_t col, bool wide,
	                                               LaneReg& lo, LaneReg& hi) {
		wave_fence();        // mask stores of earlier steps -> visible to whichever lane re-reads them
		const uint32_t ln = threadIdx.x & 63, d = ln & 15, g = ln >> 4;
		const int r = (int)row - (int)d - (g == 1 ? 1 : 0);
		const int c = (int)col - (int)d - (g == 2 ? 1 : 0);
		uint32_t v = 0, vh = 0;
		if (r >= 0 && c >= 0) {
			if (g < 3) {
				if (wide) { const uint64_t x = reinterpret_cast<const uint64_t*>(dp.mat)[dp_cell(R, (uint32_t)r, (uint32_t)c)]; v = (uint32_t)x; vh = (uint32_t)(x >> 32); }
				else v = dp.mat[dp_cell(R, (uint32_t)r, (uint32_t)c)];
			} else v = dp.masks[(uint64_t)r * cols + (uint32_t)c];
		}
		lo = v; hi = vh;
	}
	// reference window -> masks, one base per lane per pass (SwAligner::initRef, aligner_sw.cpp:155-271)
	static __device__ __forceinline__ void fetch_ref(const DevRef& ref, Work& w, uint64_t tidx, int64_t rfi, uint32_t count) {
		wave_fence();
		const uint64_t rec0 = ref_rec_find(ref, tidx, rfi);      // the window's first record: one search for the whole window
		for (uint32_t i = threadIdx.x & 63; i < count; i += 64) dev_rf()[i] = (uint8_t)(1 << ref_base_at(ref, tidx, rfi + (int64_t)i, rec0));
		wave_fence();
	}
	// a window that lies inside one N-free fragment is `count` consecutive characters of the joined text: no record search
	static __device__ __forceinline__ void fetch_ref_joined(const DevRef& ref, uint64_t jpos, uint32_t count) {
		wave_fence();
		const uint8_t* buf = uni_ptr(ref.buf);
		for (uint32_t i = threadIdx.x & 63; i < count; i += 64) { const uint64_t p = jpos + i; dev_rf()[i] = (uint8_t)(1u << ((gld(buf + (p >> 2)) >> ((p & 3) << 1)) & 3)); }
		wave_fence();
	}
	// the same window as base codes 0..4 (ungappedAlign compares characters, aligner_sw.cpp:330-380)
	static __device__ __forceinline__ void fetch_ref_codes(const DevRef& ref, uint64_t tidx, int64_t rfi, uint32_t count) {
		wave_fence();
		const uint64_t rec0 = ref_rec_find(ref, tidx, rfi);
		for (uint32_t i = threadIdx.x & 63; i < count; i += 64) dev_rf()[i] = (uint8_t)ref_base_at(ref, tidx, rfi + (int64_t)i, rec0);
		wave_fence();
	}
	// (a thin dispatcher: the fills themselves are leaf functions, see fill_ee_u8_leaf)
	static __device__ __attribute__((noinline)) int64_t dp_fill_local(const AlignParams&, Work&, bool fw_, uint32_t rows_, uint32_t cols_, uint32_t* mat_,
	                                                                  int64_t minsc_, uint32_t& lastsolcol, uint32_t& sat8) {
	wave_fence();
		const bool fw = uni((int)fw_) != 0;
		const uint32_t rows = uni(rows_), cols = uni(cols_);
		const int64_t minsc = uni(minsc_);
		uint8_t* m64 = reinterpret_cast<uint8_t*>(uni_ptr(mat_));      // one predecessor byte per cell, anti-diagonal order (pred_at)
		const int ms = minsc > 0x7fff ? 0x7fff : (int)minsc;
		if ((threadIdx.x & 63) == 0) { g_st.dp.epoch[1] = dp_RB(rows); g_st.dp.epoch[2] = 0u; }      // geometry of the matrix in the scratch header: anti-diagonal form
		int best;
		// the worker's fills leave their candidate cells in Work::cands_tmp for gather_local; the stage kernel (k_dp_fill) has no arena
		BT2_G BtCand* const emit = g_st.emit_on ? &DevPlat::work().cands_tmp[0] : (BT2_G BtCand*)nullptr;
		const uint32_t ecap = (uint32_t)kMaxCands;
		if (emit) {
			// pass 1 (workers only; the stage kernel always stores its matrix): does the window reach the minimum score at all?
			switch (dp_RB(rows)) {
				case 1: best = fill_local_leaf<1, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
				case 2: best = fill_local_leaf<2, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
				case 3: best = fill_local_leaf<3, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#ifdef BT2G_KCLASS_LR
				case 4: best = fill_local_leaf<4, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
				case 8: best = fill_local_leaf<8, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
				default: best = fill_local_leaf<16, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#else
				default: best = fill_local_leaf<4, false, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#endif
			}
			best = uni(best);
			wave_fence();
			g_hot.n_dp_cells_score += rows * cols;
			if ((int64_t)best < minsc) { lastsolcol = uni(g_st.fill_lastsol); sat8 = uni(g_st.fill_sat8); return (int64_t)best; }
			g_hot.n_dp_pass++;
		}
		if (emit) switch (dp_RB(rows)) {
			case 1: best = fill_local_leaf<1, true>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 2: best = fill_local_leaf<2, true>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 3: best = fill_local_leaf<3, true>(fw, rows, cols, m64, ms, emit, ecap); break;
#ifdef BT2G_KCLASS_LR      // (dp_RB rounds up to what is instantiated)
			case 4: best = fill_local_leaf<4, true>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 8: best = fill_local_leaf<8, true>(fw, rows, cols, m64, ms, emit, ecap); break;
			default: best = fill_local_leaf<16, true>(fw, rows, cols, m64, ms, emit, ecap); break;
#else
			default: best = fill_local_leaf<4, true>(fw, rows, cols, m64, ms, emit, ecap); break;
#endif
		} else switch (dp_RB(rows)) {
			case 1: best = fill_local_leaf<1, false>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 2: best = fill_local_leaf<2, false>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 3: best = fill_local_leaf<3, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#ifdef BT2G_KCLASS_LR
			case 4: best = fill_local_leaf<4, false>(fw, rows, cols, m64, ms, emit, ecap); break;
			case 8: best = fill_local_leaf<8, false>(fw, rows, cols, m64, ms, emit, ecap); break;
			default: best = fill_local_leaf<16, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#else
			default: best = fill_local_leaf<4, false>(fw, rows, cols, m64, ms, emit, ecap); break;
#endif
		}
		best = uni(best);
		wave_fence();
		lastsolcol = uni(g_st.fill_lastsol); sat8 = uni(g_st.fill_sat8);
		g_hot.n_dp_cells_full += rows * cols;
		return (int64_t)best;
	}
	// Candidate cells of a local fill (gatherCellsNucleotidesLocalSseU8), ordered score desc, row desc, col desc (DpBtCandidate::operator<).
	// The fill left them in Work::cands_tmp in the order it met them (fill_local_wave<R, true>); this sorts them into `cands`: a stable
	// LSD radix sort, 10 bits per pass, on  key = score : row : col  (each field as wide as this window needs), complemented so that
	// ascending passes give the descending order.  The 1024 counters of a pass are words in LDS (the last-row buffer, idle here): 16 bits
	// wide in the classes whose capacity keeps every count below 65536, 32 bits in the long-read class (kRadixCntBytes).  Per batch of 64 records: ten ballots tell every lane which lanes hold the same digit,
	// so ranks within the batch need no atomics and the scatter is stable.  An odd number of passes ends in `cands`.
	// Candidates in columns beyond lastsolcol (ncol - 1) are dropped by the first pass.
	static __device__ __attribute__((noinline)) uint32_t gather_local(const uint32_t*, BtCand* cands_, uint32_t cap_, bool, uint32_t, uint32_t rows_,
	                                                                  uint32_t ncol_, int64_t, uint32_t, uint32_t*) {
		BT2_G BtCand* const dst = (BT2_G BtCand*)uni_ptr(cands_);
		BT2_G BtCand* const tmp = &DevPlat::work().cands_tmp[0];
		const uint32_t rows = uni(rows_), ncol = uni(ncol_);
		typedef std::conditional<(kMaxCands > 65536), uint32_t, uint16_t>::type Cnt;
		uint32_t cap = uni(cap_); if (sizeof(Cnt) == 2 && cap > 65535u) cap = 65535u;
		uint32_t n = uni(g_st.n_emit);
		if (n > cap) return (uint32_t)kMaxCands + 1u;      // more cells than the lists hold: the caller flags the read
		if (n == 0) return 0;
		const uint32_t lane = threadIdx.x & 63;
		const unsigned long long lt = (1ull << lane) - 1ull;
		auto nbits = [](uint32_t v) -> uint32_t { return v ? 32u - (uint32_t)__builtin_clz(v) : 1u; };
		const uint32_t cb = nbits(ncol - 1), rb = nbits(rows - 1), sb = nbits((uint32_t)uni(g_st.emit_vmax));
		const uint32_t tot = cb + rb + sb;
		uint32_t npass = (tot + 9u) / 10u; if (!(npass & 1u)) npass++;
		const uint64_t kmask = tot >= 64u ? ~0ull : ((1ull << tot) - 1ull);
		Cnt* const cnt = reinterpret_cast<Cnt*>(dev_lastrow());       // 1024 counters (hot_tail_bytes keeps at least kRadixCntBytes there)
		uint32_t* const cnt32 = reinterpret_cast<uint32_t*>(dev_lastrow());
		for (uint32_t p = 0; p < npass; p++) {
			BT2_G BtCand* const src = (p & 1u) ? dst : tmp;
			BT2_G BtCand* const out = (p & 1u) ? tmp : dst;
			const uint32_t sh = 10u * p;
			wave_fence();
			for (uint32_t i = lane; i < kRadixCntBytes / 4u; i += 64) cnt32[i] = 0;
			wave_fence();
			auto digit_of = [&](const BtCand& c) -> uint32_t {
				const uint64_t key = (((uint64_t)(uint32_t)c.score << (rb + cb)) | ((uint64_t)c.row << cb) | (uint64_t)c.col) ^ kmask;
				return sh >= 64u ? 0u : (uint32_t)((key >> sh) & 1023ull);
			};
			// lanes holding the same digit as this lane (among the valid ones)
			auto same_digit = [&](bool valid, uint32_t d) -> unsigned long long {
				unsigned long long m = __ballot(valid);
#pragma unroll
				for (uint32_t b = 0; b < 10; b++) { const bool bit = ((d >> b) & 1u) != 0; const unsigned long long bal = __ballot(valid && bit); m &= bit ? bal : ~bal; }
				return m;
			};
			// histogram (four groups of 64 candidates per round: their loads are in flight together -- one memory round trip per 256 candidates instead of four)
			for (uint32_t base = 0; base < n; base += 256) {
				BtCand cc[4];
#pragma unroll
				for (uint32_t u = 0; u < 4; u++) { const uint32_t i = base + 64u * u + lane; cc[u].score = 0; cc[u].row = 0; cc[u].col = 0; if (i < n) cc[u] = gld(src + i); }
#pragma unroll
				for (uint32_t u = 0; u < 4; u++) {
					const uint32_t i = base + 64u * u + lane;
					if (base + 64u * u >= n) break;
					const BtCand c = cc[u];
					const bool valid = i < n && (p > 0 || (uint32_t)c.col < ncol);
					const uint32_t d = digit_of(c);
					const unsigned long long m = same_digit(valid, d);
					if (valid && (m & lt) == 0) cnt[d] = (Cnt)(cnt[d] + (uint32_t)__popcll(m));
					wave_fence();
				}
			}
			// exclusive prefix over the 1024 counters: lane l owns counters 16 l .. 16 l + 15
			uint32_t mine[16], sum = 0;
#pragma unroll
			for (uint32_t q = 0; q < 16; q++) { mine[q] = cnt[16u * lane + q]; sum += mine[q]; }
			uint32_t incl = sum;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += t; }
			uint32_t run = incl - sum;
			const uint32_t total = (uint32_t)__shfl((int)incl, 63);
			wave_fence();
#pragma unroll
			for (uint32_t q = 0; q < 16; q++) { cnt[16u * lane + q] = (Cnt)run; run += mine[q]; }
			wave_fence();
			// stable scatter (groups taken in order; loads four groups ahead as above)
			for (uint32_t base = 0; base < n; base += 256) {
				BtCand cc[4];
#pragma unroll
				for (uint32_t u = 0; u < 4; u++) { const uint32_t i = base + 64u * u + lane; cc[u].score = 0; cc[u].row = 0; cc[u].col = 0; if (i < n) cc[u] = gld(src + i); }
#pragma unroll
				for (uint32_t u = 0; u < 4; u++) {
					const uint32_t i = base + 64u * u + lane;
					if (base + 64u * u >= n) break;
					const BtCand c = cc[u];
					const bool valid = i < n && (p > 0 || (uint32_t)c.col < ncol);
					const uint32_t d = digit_of(c);
					const unsigned long long m = same_digit(valid, d);
					uint32_t s0 = 0;
					if (valid) s0 = cnt[d];
					wave_fence();
					if (valid) {
						const uint32_t rank = (uint32_t)__popcll(m & lt), grp = (uint32_t)__popcll(m);
						gst(out + s0 + rank, c);
						if (rank + 1u == grp) cnt[d] = (Cnt)(s0 + grp);
					}
					wave_fence();
				}
			}
			n = total;       // (the first pass dropped the columns beyond lastsolcol)
			if (n == 0) break;
		}
		wave_fence();
		return n;
	}
	// returns the best last-row score (de-biased)
	// (Arguments of a real call arrive in vector registers and a load through a generic reference could be a per-lane scratch access: the
	// compiler must treat both as lane-varying, and every loop bound or condition derived from them becomes exec-mask control flow and
	// vector arithmetic.  So: the parameter block and the scratch descriptor are read from their LDS objects BY NAME, the scalar arguments
	// go through v_readfirstlane once.)  A thin dispatcher: the fills themselves are leaf functions (fill_ee_u8_leaf).
	// Does a 16-bit end-to-end problem of this shape go through the band fill (predecessor bytes; the backtrace then takes the 8-bit fill's path,
	// with the 16-bit kernel's RNG protocol) or -- a band of more than 2 048 diagonals -- through the anti-diagonal fill of the whole rectangle?
	static __device__ __forceinline__ bool ee_wide_band(uint32_t rows, uint32_t cols, int64_t minsc) {
		EeBand band;
		if (!ee_band(uni(g_P.rfgapo), uni(g_P.rfgape), uni(rows), uni(cols), uni(minsc), band)) return true;      // (no cell can reach minsc: no matrix either way)
		return ee_band_rp(band.nd) != 0u;
	}
	template <bool PRED, bool W16>
	static __device__ __forceinline__ int band_leaf(uint32_t rp, bool fw, uint32_t rows, uint32_t cols, int lo, int thr, uint8_t* pm) {
		switch (rp) {
			case 1: return fill_ee_u8_leaf<1, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 2: return fill_ee_u8_leaf<2, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 3: return fill_ee_u8_leaf<3, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 4: return fill_ee_u8_leaf<4, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 6: return fill_ee_u8_leaf<6, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 8: return fill_ee_u8_leaf<8, PRED, W16>(fw, rows, cols, lo, thr, pm);
			case 12: return fill_ee_u8_leaf<12, PRED, W16>(fw, rows, cols, lo, thr, pm);
			default: return fill_ee_u8_leaf<16, PRED, W16>(fw, rows, cols, lo, thr, pm);
		}
	}
	static __device__ __attribute__((noinline)) int64_t dp_fill_ee(const AlignParams&, Work&, bool fw_, uint32_t rows_, uint32_t cols_, const DpScratch&, bool wide_, int64_t minsc_) {
		wave_fence();     // w.rf / read written by the scalar code -> visible to every lane
		const AlignParams& P = g_P;
		const DpScratch& dp = g_st.dp;
		const bool fw = uni((int)fw_) != 0, wide = uni((int)wide_) != 0;
		const uint32_t rows = uni(rows_), cols = uni(cols_);
		const int64_t minsc = uni(minsc_);
		uint32_t* mat = uni_ptr(dp.mat);
		int best;
		if (!wide || (ee_wide_band(rows, cols, minsc) && !uni(g_st.wide_cells))) {
			uint8_t* pm = reinterpret_cast<uint8_t*>(mat);
			const int bias = wide ? 0xffff : 0xff;
			EeBand band;
			if (!ee_band(P.rfgapo, P.rfgape, rows, cols, minsc, band)) return -(int64_t)bias;      // no cell can lie on an alignment that reaches minsc
			const uint32_t rp = ee_band_rp(band.nd);
			if (rp == 0) return INT64_MIN;
			const int lo = band.lo;
			const int thr = (int)(minsc + bias);      // biased score an alignment must keep (>= 1: the 8-bit kernel is only used while minsc >= -254, and no read is long enough for -65 534)
			// pass 1: can any end-to-end alignment in this window reach the minimum score at all?  A lower bound on the best score costs
			// almost nothing: the gap-free alignment along the window's middle diagonal (the seed's own diagonal unless the window was
			// trimmed at a reference end) is one of the alignments the fill maximises over -- it lies inside the band, and while its score
			// stays >= minsc nothing on it saturates.  When it already reaches minsc (a read without an indel at its true locus, or
			// at a close copy: most windows that succeed at all) the score-only pass has nothing left to decide and is skipped.
			bool skip1 = false;
			if (cols >= rows) {
				const uint32_t dd = (cols - rows) >> 1;
				int pen = 0;
				for (uint32_t i = threadIdx.x & 63; i < rows; i += 64) {
					const int c = rd_char(g_hot, g_hot.len, fw, i);
					const int rfm = dev_rf()[i + dd];
					if (c > 3 || rfm > 15) pen += P.n_pen;
					else if (!((rfm >> c) & 1)) { const int q = rd_qual(g_hot, g_hot.len, fw, i) - 33; pen += mm_penalty(P, q < 0 ? 0 : q); }
				}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) pen += __shfl_xor(pen, o);
				skip1 = (int64_t)(-uni(pen)) >= minsc;
			}
			if (skip1) best = thr;      // (any value that passes the test below: the matrix pass computes the real one)
			else best = wide ? band_leaf<false, true>(rp, fw, rows, cols, lo, thr, pm) : band_leaf<false, false>(rp, fw, rows, cols, lo, thr, pm);
			best = uni(best);
			wave_fence();
			const uint32_t rows_done = skip1 ? 0u : uni(g_st.fill_rows_done);
			g_hot.n_dp_cells_score += rows_done * band.nd;
			if ((int64_t)best - bias < minsc) { wave_fence(); return (int64_t)best - bias; }
			g_hot.n_dp_cells_full += rows * band.nd; g_hot.n_dp_pass++;
			// pass 2: the matrix of predecessor bits (same scores, so `best` is unchanged)
			if ((threadIdx.x & 63) == 0) { dp.epoch[1] = (uint32_t)lo; dp.epoch[2] = 128u * rp; }
			best = wide ? band_leaf<true, true>(rp, fw, rows, cols, lo, thr, pm) : band_leaf<true, false>(rp, fw, rows, cols, lo, thr, pm);
			best = uni(best) - bias;
		} else {
			uint64_t* m64 = reinterpret_cast<uint64_t*>(mat);
			switch (dp_R(rows)) {
				case 1: best = fill_ee_i16_leaf<1>(fw, rows, cols, m64); break;
				case 2: best = fill_ee_i16_leaf<2>(fw, rows, cols, m64); break;
				case 3: best = fill_ee_i16_leaf<3>(fw, rows, cols, m64); break;
				case 4: best = fill_ee_i16_leaf<4>(fw, rows, cols, m64); break;
				case 5: best = fill_ee_i16_leaf<5>(fw, rows, cols, m64); break;
				case 6: best = fill_ee_i16_leaf<6>(fw, rows, cols, m64); break;
				case 7: best = fill_ee_i16_leaf<7>(fw, rows, cols, m64); break;
#ifdef BT2G_KCLASS_LR
				case 8: best = fill_ee_i16_leaf<8>(fw, rows, cols, m64); break;
				case 16: best = fill_ee_i16_leaf_big<16>(fw, rows, cols, m64); break;      // (dp_R rounds up to what is instantiated)
				case 24: best = fill_ee_i16_leaf_big<24>(fw, rows, cols, m64); break;
				default: best = fill_ee_i16_leaf_big<32>(fw, rows, cols, m64); break;
#else
				default: best = fill_ee_i16_leaf<8>(fw, rows, cols, m64); break;
#endif
			}
			best = uni(best) - 0x7fff;
			g_hot.n_dp_cells_full += rows * cols;
		}
		wave_fence();     // matrix written lane-parallel -> visible to the scalar backtrace
		return (int64_t)best;
	}
};

// one DP scratch inside a wave's arena: [matrix][16-bit masks][256 B header holding the epoch][32-bit epoch-tagged masks]
__device__ __forceinline__ uint8_t* carve_scratch(DpScratch& dp, uint8_t* p, uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes) {
	dp.mat = (BT2_G uint32_t*)p; p += mat_bytes;
	dp.masks = (BT2_G uint16_t*)p; p += mask_bytes;
	dp.epoch = (BT2_G uint32_t*)p;
	dp.pmask = (BT2_G uint32_t*)(p + 256);
	dp.pmask_words = (uint32_t)((pmask_bytes - 256) / 4);
	return p + pmask_bytes;
}

// per-read parameters the host derives (seed length 1..32 as -L allows, positive seed interval); anything else would index
// past the seed tables
__device__ __forceinline__ bool read_params_ok(const ReadParams& rp) { return rp.seedlen >= 1 && rp.seedlen <= 32 && rp.interval >= 1 && rp.nceil >= 0; }

// Reads (pairs) a persistent wave takes from the queue per atomic.
#ifndef BT2G_QCHUNK
#define BT2G_QCHUNK 4
#endif
constexpr unsigned int kQueueChunk = BT2G_QCHUNK, kQueueChunkPairs = BT2G_QCHUNK > 2 ? 2 : BT2G_QCHUNK;
#ifndef BT2G_WAVES_PER_EU
#define BT2G_WAVES_PER_EU 2
#endif
#ifndef BT2G_NUM_VGPR
#define BT2G_NUM_VGPR 128      // architectural vector registers of the worker kernels (512 / 4 waves per SIMD)
#endif
template <typename TOff>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BT2G_WAVES_PER_EU, BT2G_WAVES_PER_EU), amdgpu_num_vgpr(BT2G_NUM_VGPR)))
k_align_reads(DevIndex<TOff> ix, AlignParams P, bt2g_reads rd, const ReadParams* __restrict__ rparams,
              uint8_t* __restrict__ results, uint64_t result_stride, uint8_t* __restrict__ arena, uint64_t arena_stride,
              uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, unsigned int* __restrict__ next_read, unsigned long long* __restrict__ prof,
              PreComp pre, uint32_t max_read_len, uint32_t max_cols, uint32_t rt_bytes, uint32_t dyn_extra) {
	const int lane = threadIdx.x & 63;
	uint8_t* base = arena + (uint64_t)blockIdx.x * arena_stride;
	BT2_G Work& w = *(BT2_G Work*)base;
	DpScratch dp;
	carve_scratch(dp, base + ((sizeof(Work) + 255) & ~(uint64_t)255), mat_bytes, mask_bytes, pmask_bytes);
	__shared__ alignas(16) unsigned char s_al[sizeof(Aligner<TOff, DevPlat>)];
	*reinterpret_cast<DevIndex<TOff>*>(g_ix_raw) = ix; g_P = P; g_pre = pre;
	g_st.max_cols = max_cols; g_st.rf_at = lds_addr_of_tail(); g_st.ned_at = g_st.rf_at + hot_tail_off(max_cols);
	g_st.rt_at = g_st.rf_at + hot_tail_bytes(max_cols, P.match_bonus > 0); g_st.rt_bytes = rt_bytes; g_st.rt_cur = 0; g_st.wide_cells = 0;
	g_st.dyn_bytes = hot_tail_bytes(max_cols, P.match_bonus > 0) + rt_bytes + dyn_extra;
	wave_fence();
	// The queue head is ONE word that every wave of the device adds to: the waves take reads kQueueChunk at a time (see the note at kQueueChunk)
	unsigned int r_next = 0, r_end = 0;
	for (;;) {
		if (r_next == r_end) {
			unsigned int r0 = 0;
			if (lane == 0) r0 = atomicAdd(next_read, kQueueChunk);
			r_next = (unsigned int)__builtin_amdgcn_readfirstlane((int)r0); r_end = r_next + kQueueChunk;
		}
		const unsigned int r = r_next++;
		if (r >= rd.n_reads) break;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		BT2_G ReadResult& out = *(BT2_G ReadResult*)(results + (uint64_t)r * result_stride);
		if (len > max_read_len || !read_params_ok(rparams[r])) {      // the DP scratch of this launch is sized for max_read_len rows
			if (lane == 0) { out.status = ERR_OVERFLOW; out.aligned = 0; out.nreport = 0; out.nalns = 0; out.filt = (uint8_t)rparams[r].filt; out.maxed = 0; out.has_secbest = 0; }
			continue;
		}
		// stage the read into the work area (lane-parallel copy)
		wave_fence();
		g_hot.len = len;
		for (uint32_t i = lane; i < len; i += 64) { g_hot.seq[i] = rd.d_seq[o0 + i]; g_hot.qual[i] = rd.d_qual[o0 + i]; }
		wave_fence();
		// The control state (`this`, the parameter blocks) is kept in LDS: the worker's member functions
		// are real calls, and anything they reach through a pointer would otherwise be a private-memory load.
		g_rp = rparams[r];
		wave_fence();
		Aligner<TOff, DevPlat>& al = *new (s_al) Aligner<TOff, DevPlat>(w, dp, r);
		wave_fence();
		al.run(out);
		wave_fence();
		if (lane == 0 && prof) {
			for (int i = 0; i < 8; i++) atomicAdd(&prof[i], (unsigned long long)g_hot.t_phase[i]);
			for (int i = 8; i < 22; i++) atomicAdd(&prof[i + 2], (unsigned long long)g_hot.t_phase[i]);
			atomicAdd(&prof[8], (unsigned long long)g_hot.n_sides);
			atomicAdd(&prof[9], 1ull);
			atomicAdd(&prof[24], (unsigned long long)g_hot.n_dp_cells_score); atomicAdd(&prof[25], (unsigned long long)g_hot.n_dp_cells_full); atomicAdd(&prof[26], (unsigned long long)(g_hot.n_dp_pass & 0xffffu) | ((unsigned long long)(g_hot.n_dp_pass >> 16) << 32)); for (int i = 0; i < 5; i++) atomicAdd(&prof[27 + i], (unsigned long long)g_hot.t_bt[i]);
		}
	}
}

#ifndef BT2G_NO_PAIRS
// Paired-end flavour: one wavefront per pair (reads 2p and 2p+1, result records 2p and 2p+1).  A separate kernel so
// that the unpaired kernel's register allocation and code layout do not carry the pair logic.
template <typename TOff>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BT2G_WAVES_PER_EU, BT2G_WAVES_PER_EU), amdgpu_num_vgpr(BT2G_NUM_VGPR)))
k_align_pairs(DevIndex<TOff> ix, AlignParams P, bt2g_reads rd, const ReadParams* __restrict__ rparams,
              uint8_t* __restrict__ results, uint64_t result_stride, uint8_t* __restrict__ arena, uint64_t arena_stride,
              uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, unsigned int* __restrict__ next_read, unsigned long long* __restrict__ prof,
              PreComp pre, uint32_t max_read_len, uint32_t max_cols, uint32_t rt_bytes, uint32_t dyn_extra) {
	const int lane = threadIdx.x & 63;
	uint8_t* base = arena + (uint64_t)blockIdx.x * arena_stride;
	BT2_G Work& w = *(BT2_G Work*)base;
	DpScratch dp, dp2;
	carve_scratch(dp2, carve_scratch(dp, base + ((sizeof(Work) + 255) & ~(uint64_t)255), mat_bytes, mask_bytes, pmask_bytes), mat_bytes, mask_bytes, pmask_bytes);
	__shared__ alignas(16) unsigned char s_al[sizeof(Aligner<TOff, DevPlat>)];
	*reinterpret_cast<DevIndex<TOff>*>(g_ix_raw) = ix; g_P = P; g_pre = pre;
	g_st.max_cols = max_cols; g_st.rf_at = lds_addr_of_tail(); g_st.ned_at = g_st.rf_at + hot_tail_off(max_cols);
	g_st.rt_at = 0; g_st.rt_bytes = 0; g_st.rt_cur = 0; g_st.wide_cells = 0;      // (two matrices in flight: their marks stay in the arena)
	g_st.dyn_bytes = hot_tail_bytes(max_cols, P.match_bonus > 0) + dyn_extra;
	(void)rt_bytes;
	wave_fence();
	const unsigned int n_pairs = rd.n_reads / 2;
	unsigned int r_next = 0, r_end = 0;
	for (;;) {
		if (r_next == r_end) {
			unsigned int r0 = 0;
			if (lane == 0) r0 = atomicAdd(next_read, kQueueChunkPairs);
			r_next = (unsigned int)__builtin_amdgcn_readfirstlane((int)r0); r_end = r_next + kQueueChunkPairs;
		}
		const unsigned int r = r_next++;
		if (r >= n_pairs) break;
		const uint64_t o0 = rd.d_off[2 * r], o1 = rd.d_off[2 * r + 1], o2 = rd.d_off[2 * r + 2];
		const uint32_t len0 = (uint32_t)(o1 - o0), len1 = (uint32_t)(o2 - o1);
		BT2_G ReadResult& out0 = *(BT2_G ReadResult*)(results + (uint64_t)(2 * r) * result_stride);
		BT2_G ReadResult& out1 = *(BT2_G ReadResult*)(results + (uint64_t)(2 * r + 1) * result_stride);
		if (len0 > max_read_len || len1 > max_read_len || !read_params_ok(rparams[2 * r]) || !read_params_ok(rparams[2 * r + 1])) {
			if (lane == 0) {
				BT2_G ReadResult* o[2] = {&out0, &out1};
				for (int m = 0; m < 2; m++) { o[m]->status = ERR_OVERFLOW; o[m]->aligned = 0; o[m]->nreport = 0; o[m]->nalns = 0; o[m]->filt = (uint8_t)rparams[2 * r + m].filt; o[m]->maxed = 0; o[m]->has_secbest = 0; o[m]->pair_type = 0; o[m]->pair_flags = 0; }
			}
			continue;
		}
		wave_fence();
		g_rp = rparams[2 * r];
		wave_fence();
		Aligner<TOff, DevPlat>& al = *new (s_al) Aligner<TOff, DevPlat>(w, dp, 2 * r);
		g_st.dp_main = dp; g_st.dp_opp = dp2;
		g_st.pe_seq[0] = (const BT2_G uint8_t*)(rd.d_seq + o0); g_st.pe_qual[0] = (const BT2_G uint8_t*)(rd.d_qual + o0); g_st.pe_len[0] = len0;
		g_st.pe_seq[1] = (const BT2_G uint8_t*)(rd.d_seq + o1); g_st.pe_qual[1] = (const BT2_G uint8_t*)(rd.d_qual + o1); g_st.pe_len[1] = len1;
		g_st.pe_rp[0] = rparams[2 * r]; g_st.pe_rp[1] = rparams[2 * r + 1];
		g_st.pe_pair = r;
		wave_fence();
		al.run_pair(out0, out1);
		wave_fence();
		if (lane == 0 && prof) {
			for (int i = 0; i < 8; i++) atomicAdd(&prof[i], (unsigned long long)g_hot.t_phase[i]);
			for (int i = 8; i < 22; i++) atomicAdd(&prof[i + 2], (unsigned long long)g_hot.t_phase[i]);
			atomicAdd(&prof[8], (unsigned long long)g_hot.n_sides);
			atomicAdd(&prof[9], 2ull);
			atomicAdd(&prof[24], (unsigned long long)g_hot.n_dp_cells_score); atomicAdd(&prof[25], (unsigned long long)g_hot.n_dp_cells_full); atomicAdd(&prof[26], (unsigned long long)(g_hot.n_dp_pass & 0xffffu) | ((unsigned long long)(g_hot.n_dp_pass >> 16) << 32)); for (int i = 0; i < 5; i++) atomicAdd(&prof[27 + i], (unsigned long long)g_hot.t_bt[i]);
		}
	}
}

#endif

template <typename TOff>
hipError_t launch_align(const DevIndex<TOff>& ix, const AlignParams& P, const bt2g_reads& rd, const ReadParams* d_rparams,
                        uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride,
                        uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof,
                        const PreComp& pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st) {
	if (rd.n_reads == 0) return hipSuccess;
	hipError_t e = hipMemsetAsync(d_next, 0, sizeof(unsigned int), st);
	if (e != hipSuccess) return e;
	const uint32_t tail = hot_tail_bytes(max_cols, P.match_bonus > 0);      // dynamic LDS: the per-column tail of the hot state
	// What is left of the wave's share of LDS (lds_per_wave: what keeps this class's waves per CU resident; 512 bytes are left alone) is launched as
	// dynamic LDS as well: the row sampler's hash tables live there between DP windows and hold the more entries the more there is (DevPlat::sh_begin).
	auto spare = [&](const void* kfn, uint32_t dyn_used) -> uint32_t {
		hipFuncAttributes fa2{};
		if (lds_per_wave == 0 || hipFuncGetAttributes(&fa2, kfn) != hipSuccess) return 0u;
		const uint64_t used = (uint64_t)fa2.sharedSizeBytes + dyn_used + 512u;
		return used < lds_per_wave ? (uint32_t)((lds_per_wave - used) & ~15ull) : 0u;
	};
	if (P.paired) {
#ifdef BT2G_NO_PAIRS
		return hipErrorInvalidValue;      // (this class holds no pair state: bt2g_align_batch never sends it a batch of pairs)
#else
		const uint32_t extra = spare(reinterpret_cast<const void*>(&k_align_pairs<TOff>), tail);
		hipLaunchKernelGGL(k_align_pairs<TOff>, dim3(n_waves), dim3(64), tail + extra, st, ix, P, rd, d_rparams, d_results, result_stride,
		                   d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, d_next, d_prof, pre, max_read_len, max_cols, 0u, extra);
#endif
	} else {
		// ... and to the on-chip backtrace state of end-to-end batches: the reportedThrough plane of a band matrix of the longest read at the
		// narrowest band (16 bytes per row -- a wider band's marks stay in the arena, DevPlat::rt_begin decides per matrix).
		uint32_t rt_bytes = 0;
		hipFuncAttributes fa{};
		static const bool rt_off = getenv("BT2G_RT_LDS") && atoi(getenv("BT2G_RT_LDS")) == 0;          // measurement knob: marks in the arena, as before round 5
		if (!rt_off && P.match_bonus == 0 && lds_per_wave != 0 && hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_align_reads<TOff>)) == hipSuccess) {
			const uint64_t used = (uint64_t)fa.sharedSizeBytes + tail;
			const uint32_t want = ((max_read_len ? max_read_len : 1u) * 16u + 15u) & ~15u;
			if (used + want <= lds_per_wave) rt_bytes = want;
		}
		const uint32_t extra = spare(reinterpret_cast<const void*>(&k_align_reads<TOff>), tail + rt_bytes);
		static const bool dbg_occ = getenv("BT2G_DEBUG_OCC") != nullptr;
		if (dbg_occ) {
			int nb = 0;
			(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_align_reads<TOff>), 64, tail + rt_bytes + extra);
			fprintf(stderr, "[bt2g] k_align_reads: %d waves per CU resident (LDS %u static + %u tail + %u marks + %u spare; budget %u per wave), %u waves launched\n",
			        nb, (unsigned)fa.sharedSizeBytes, tail, rt_bytes, extra, lds_per_wave, n_waves);
		}
		hipLaunchKernelGGL(k_align_reads<TOff>, dim3(n_waves), dim3(64), tail + rt_bytes + extra, st, ix, P, rd, d_rparams, d_results, result_stride,
		                   d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, d_next, d_prof, pre, max_read_len, max_cols, rt_bytes, extra);
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The fills as a stage (bt2g_dp_fill, include/bt2g.h): every problem goes through DevPlat::dp_fill_ee / dp_fill_local -- the
// functions the worker calls -- on a wave's private scratch, and what they leave behind is copied to the problem's output block.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BT2G_WAVES_PER_EU, BT2G_WAVES_PER_EU), amdgpu_num_vgpr(BT2G_NUM_VGPR)))
k_dp_fill(AlignParams P, const bt2g_dp_problem* __restrict__ probs, uint32_t n, const uint8_t* __restrict__ d_rd, const uint8_t* __restrict__ d_qu,
          const uint8_t* __restrict__ d_rf, uint8_t* __restrict__ d_out, uint8_t* __restrict__ scratch, uint64_t scratch_stride,
          uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t max_cols) {
	const uint32_t lane = threadIdx.x & 63;
	g_P = P;
	g_st.max_cols = max_cols; g_st.rf_at = lds_addr_of_tail(); g_st.ned_at = g_st.rf_at + hot_tail_off(max_cols);
	g_st.rt_at = 0; g_st.rt_bytes = 0; g_st.rt_cur = 0; g_st.dyn_bytes = 0; g_st.wide_cells = 0;
	DpScratch dp;
	carve_scratch(dp, scratch + (uint64_t)blockIdx.x * scratch_stride, mat_bytes, mask_bytes, pmask_bytes);
	g_st.dp = dp; g_st.wp = (BT2_G Work*)scratch; g_st.emit_on = 0;      // (the fills do not touch the work area)
	wave_fence();
	for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
		const bt2g_dp_problem pr = probs[p];
		const uint32_t rows = pr.rows, cols = pr.cols;
		BT2_G bt2g_dp_out* out = (BT2_G bt2g_dp_out*)(d_out + pr.out_off);
		wave_fence();
		g_hot.len = rows;
		for (uint32_t i = lane; i < rows; i += 64) { g_hot.seq[i] = d_rd[pr.rd_off + i]; g_hot.qual[i] = d_qu[pr.rd_off + i]; }
		for (uint32_t j = lane; j < cols + 1; j += 64) dev_rf()[j] = d_rf[pr.rf_off + j];
		g_hot.n_dp_cells_score = g_hot.n_dp_cells_full = g_hot.n_dp_pass = 0;
		wave_fence();
		int64_t best;
		uint32_t lastsolcol = 0, sat8 = 0;
		if (pr.kind == BT2G_DP_LOCAL) best = DevPlat::dp_fill_local(g_P, *g_st.wp_generic(), true, rows, cols, (uint32_t*)dp.mat, (int64_t)pr.minsc, lastsolcol, sat8);
		else {
			// BT2G_DP_EE_I16: the anti-diagonal form with its H / E / F cells; BT2G_DP_EE_I16_BAND: the 16-bit arithmetic on the band (what the worker uses wherever the band fits)
			const bool band_form = pr.kind == BT2G_DP_EE_U8 || pr.kind == BT2G_DP_EE_I16_BAND;
			if ((threadIdx.x & 63) == 0) g_st.wide_cells = pr.kind == BT2G_DP_EE_I16 ? 1u : 0u;
			wave_fence();
			best = (pr.kind == BT2G_DP_EE_I16_BAND && !DevPlat::ee_wide_band(rows, cols, (int64_t)pr.minsc)) ? INT64_MIN
			     : DevPlat::dp_fill_ee(g_P, *g_st.wp_generic(), true, rows, cols, dp, pr.kind != BT2G_DP_EE_U8, (int64_t)pr.minsc);
			(void)band_form;
		}
		wave_fence();
		const bool band_kind = pr.kind == BT2G_DP_EE_U8 || pr.kind == BT2G_DP_EE_I16_BAND;
		const bool has_mat = !band_kind || (best != INT64_MIN && best >= (int64_t)pr.minsc);
		const int32_t band_lo = (int32_t)dp.epoch[1];
		const uint32_t band_w = dp.epoch[2];
		if (lane == 0) {
			out->best = best; out->lastsolcol = lastsolcol; out->sat8 = sat8; out->pad = 0;
			out->band_lo = band_kind && has_mat ? band_lo : 0; out->band_w = band_kind && has_mat ? band_w : 0;
			out->has_matrix = has_mat ? 1u : 0u;
		}
		BT2_G uint8_t* body = (BT2_G uint8_t*)(d_out + pr.out_off + sizeof(bt2g_dp_out));
		if (band_kind) {
			const uint32_t c4 = (cols + 3u) & ~3u;
			BT2_G int16_t* lr = (BT2_G int16_t*)body;
			for (uint32_t j = lane; j < c4; j += 64) lr[j] = (has_mat && j < cols) ? dev_lastrow()[j] : (int16_t)(pr.kind == BT2G_DP_EE_U8 ? -0xff : -32768);
			if (has_mat) {
				BT2_G uint8_t* pm = body + (uint64_t)c4 * 2;
				const BT2_G uint8_t* src = (const BT2_G uint8_t*)dp.mat;
				for (uint64_t k = lane; k < (uint64_t)rows * band_w; k += 64) pm[k] = src[k];
			}
		} else if (pr.kind == BT2G_DP_LOCAL) {
			// the predecessor bytes of the rectangle, row-major (the fill keeps them in anti-diagonal order: pred_at)
			const uint32_t RB = dp_RB(rows);
			const BT2_G uint8_t* src = (const BT2_G uint8_t*)dp.mat;
			const uint64_t ncell = (uint64_t)rows * cols;
			for (uint64_t k = lane; k < ncell; k += 64) body[k] = src[dp_cell_pk(RB, (uint32_t)(k / cols), (uint32_t)(k % cols))];
		} else {
			const uint32_t R = dp_R(rows);
			const BT2_G uint64_t* m64 = (const BT2_G uint64_t*)dp.mat;
			BT2_G int32_t* H = (BT2_G int32_t*)body;
			const uint64_t ncell = (uint64_t)rows * cols;
			for (uint64_t k = lane; k < ncell; k += 64) {
				const uint32_t i = (uint32_t)(k / cols), j = (uint32_t)(k % cols);
				const uint64_t c = m64[dp_cell(R, i, j)];
				const uint32_t h = (uint32_t)(c & 0xffff), e = (uint32_t)((c >> 16) & 0xffff), f = (uint32_t)((c >> 32) & 0xffff);
				if (pr.kind == BT2G_DP_EE_I16) { H[k] = (int32_t)(int16_t)h; H[ncell + k] = (int32_t)(int16_t)e; H[2 * ncell + k] = (int32_t)(int16_t)f; }
				else { H[k] = (int32_t)h; H[ncell + k] = (int32_t)e; H[2 * ncell + k] = (int32_t)f; }
			}
		}
		wave_fence();
	}
}

hipError_t launch_dp_fill(const AlignParams& P, const bt2g_dp_problem* d_probs, uint32_t n, const uint8_t* d_rd, const uint8_t* d_qu, const uint8_t* d_rf,
                          uint8_t* d_out, uint8_t* d_scratch, uint64_t scratch_stride, uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes,
                          uint32_t n_waves, uint32_t max_cols, hipStream_t st) {
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_dp_fill, dim3(n < n_waves ? n : n_waves), dim3(64), hot_tail_bytes(max_cols), st, P, d_probs, n, d_rd, d_qu, d_rf, d_out, d_scratch, scratch_stride, mat_bytes, mask_bytes, pmask_bytes, max_cols);
	return hipGetLastError();
}

void align_scratch_sizes(uint32_t max_len, bool paired, uint32_t maxhalf, uint32_t max_cols, uint64_t& mat_bytes, uint64_t& mask_bytes, uint64_t& pmask_bytes, uint64_t& arena_stride) {
	const uint32_t rows = max_len ? max_len : 1;
	const uint32_t R = dp_R(rows);
	// unpaired: seed-extension windows only (rows + 4 * min(gaps, maxhalf) columns, dp_framer.cpp:81-129; the framer flags windows
	// past the launch's column capacity); paired: opposite-mate windows up to that capacity, and a second matrix for them
	uint32_t cols = paired ? max_cols + 4 : rows + 4 * maxhalf + 1 + 4;
	if (cols > max_cols + 4) cols = max_cols + 4;
	const uint32_t lanes = (rows + R - 1) / R;
	// packed cells (16-bit end-to-end, local): 8 B per cell, wavefront-major; pred format (8-bit end-to-end): 1 B per cell of the band
	mat_bytes = (((uint64_t)cols + lanes) * R * 64 * 8 + 255) & ~(uint64_t)255;
	const uint64_t pred_bytes = (pred_cells(rows, cols) + 255) & ~(uint64_t)255;
	if (pred_bytes > mat_bytes) mat_bytes = pred_bytes;
	mask_bytes = ((uint64_t)rows * cols * 2 + 255) & ~(uint64_t)255;
#ifdef BT2G_KCLASS_W5
	// The short-read class fills end-to-end windows of reads of at most 256 bp: always on the band (8- or 16-bit arithmetic; a band of a problem
	// this size cannot exceed 2 048 diagonals), one predecessor byte per cell -- no packed cells, no 16-bit mask plane.
	mat_bytes = pred_bytes; mask_bytes = 256;
#endif
	// masks of the pred formats: the band form of the widest band, or the anti-diagonal form of the local fill (one word per matrix byte)
	const uint64_t wf_cells = ((uint64_t)cols + 128) * dp_RB(rows) * 128;      // (dp_cell_pk: at most 128 blocks of at most dp_RB(longest read) rows)
	const uint64_t mask_cells = pred_cells(rows, cols) > wf_cells ? pred_cells(rows, cols) : wf_cells;
	pmask_bytes = 256 + ((mask_cells * 4 + 255) & ~(uint64_t)255);
	arena_stride = ((sizeof(Work) + 255) & ~(uint64_t)255) + (paired ? 2 : 1) * (mat_bytes + mask_bytes + pmask_bytes);
	arena_stride = (arena_stride + 4095) & ~(uint64_t)4095;
}

uint64_t align_work_bytes() { return sizeof(Work); }
uint32_t align_waves_per_cu() { return 4u * BT2G_WAVES_PER_EU; }

template hipError_t launch_align<uint32_t>(const DevIndex<uint32_t>&, const AlignParams&, const bt2g_reads&, const ReadParams*, uint8_t*, uint64_t, uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t, unsigned int*, unsigned long long*, const PreComp&, uint32_t, uint32_t, uint32_t, hipStream_t);
template hipError_t launch_align<uint64_t>(const DevIndex<uint64_t>&, const AlignParams&, const bt2g_reads&, const ReadParams*, uint8_t*, uint64_t, uint8_t*, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t, unsigned int*, unsigned long long*, const PreComp&, uint32_t, uint32_t, uint32_t, hipStream_t);

} // namespace bt2g

// ---------------------------------------------------------------------------------------------------------------------
// A second compilation of this file is a second OCCUPANCY CLASS of the worker (Makefile: bt2g_align_kernel_w5.o, built with
// -DBT2G_WAVES_PER_EU=5 -DBT2G_NUM_VGPR=96 -Dbt2g=bt2g_w5 -DBT2G_KCLASS_W5): the register budget of the kernels and of every device function they call
// is a property of the translation unit, so the same source is compiled again under another namespace name and reached through the two
// plain-C entry points below (the index descriptor, parameter blocks and pre-computation tables have the same layout in both: same headers).
// 49 % of a wave's cycles are spent waiting on a counter and 4 waves per SIMD cannot cover that; with 96 registers and the small
// dynamic LDS of an end-to-end batch, 5 waves per SIMD are resident.
// (A third compilation, bt2g_align_kernel_bk.o: -Dbt2g=bt2g_bk -DBT2G_KCLASS_BK -DBT2G_CLASS_BIG_K, is the many-alignments class: -k above 64 and -a.)
#define BT2G_CLASS_ENTRIES(PFX) \
extern "C" hipError_t PFX##_launch_align(int off_size, const void* ix, const bt2g_align_params* P, const bt2g_reads* rd, const bt2g_read_params* d_rparams, \
                                         uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride, \
                                         uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof, \
                                         const void* pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st) { \
	using namespace bt2g; \
	const PreComp& pc = *reinterpret_cast<const PreComp*>(pre); \
	return off_size == 4 \
		? launch_align(*reinterpret_cast<const DevIndex<uint32_t>*>(ix), *P, *rd, d_rparams, d_results, result_stride, d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, n_waves, d_next, d_prof, pc, max_read_len, max_cols, lds_per_wave, st) \
		: launch_align(*reinterpret_cast<const DevIndex<uint64_t>*>(ix), *P, *rd, d_rparams, d_results, result_stride, d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, n_waves, d_next, d_prof, pc, max_read_len, max_cols, lds_per_wave, st); \
} \
extern "C" uint32_t PFX##_waves_per_cu(void) { return bt2g::align_waves_per_cu(); } \
/* static LDS of the unpaired worker kernels of this class (the larger of the two index widths); 0xffffffff if the runtime will not say */ \
extern "C" uint32_t PFX##_static_lds(void) { \
	hipFuncAttributes a32, a64; \
	if (hipFuncGetAttributes(&a32, reinterpret_cast<const void*>(&bt2g::k_align_reads<uint32_t>)) != hipSuccess) return 0xffffffffu; \
	if (hipFuncGetAttributes(&a64, reinterpret_cast<const void*>(&bt2g::k_align_reads<uint64_t>)) != hipSuccess) return 0xffffffffu; \
	return (uint32_t)(a32.sharedSizeBytes > a64.sharedSizeBytes ? a32.sharedSizeBytes : a64.sharedSizeBytes); \
} \
extern "C" uint64_t PFX##_work_bytes(void) { return bt2g::align_work_bytes(); } \
/* what the class holds: longest read, seed positions per strand, alignments per read */ \
extern "C" uint32_t PFX##_max_len(void) { return (uint32_t)bt2g::kMaxLen; } \
extern "C" uint32_t PFX##_max_offs(void) { return (uint32_t)bt2g::kMaxOffs; } \
extern "C" uint32_t PFX##_max_alns(void) { return (uint32_t)bt2g::kMaxAlns; } \
/* the arena of one wave of this class (its Work is its own) */ \
extern "C" void PFX##_scratch_sizes(uint32_t max_len, int paired, uint32_t maxhalf, uint32_t max_cols, uint64_t* mat_bytes, uint64_t* mask_bytes, uint64_t* pmask_bytes, uint64_t* arena_stride) { \
	bt2g::align_scratch_sizes(max_len, paired != 0, maxhalf, max_cols, *mat_bytes, *mask_bytes, *pmask_bytes, *arena_stride); \
}
#ifdef BT2G_KCLASS_W5
BT2G_CLASS_ENTRIES(bt2g_w5)
#endif
#ifdef BT2G_KCLASS_BK
BT2G_CLASS_ENTRIES(bt2g_bk)
#endif
#ifdef BT2G_KCLASS_LR      // the long-read class: reads of 513 ... 1 999 bp (Makefile: bt2g_align_kernel_lr.o)
BT2G_CLASS_ENTRIES(bt2g_lr)
#endif

