// bt2g_kernels.hip -- stage kernels of the multiseed hot path for gfx950 (CDNA4).
//
//   k_exact_sweep        one lane per (read, strand): SeedAligner::exactSweep  (aligner_seed.cpp:856-970)
//   k_seed_search_exact  one lane per (read, strand, seed): startSearchSeedBi + searchSeedBi for
//                        SEED_TYPE_EXACT seeds (aligner_seed.cpp:1638-2037)
//   k_resolve_offsets    one lane per SA row: Ebwt::getOffset + joinedToTextOff (bt2_idx.cpp:54-171)
//   (the DP fills live in bt2g_align_kernel.hip: the worker's own device functions, also reachable as a stage through bt2g_dp_fill)
//
// The FM kernels are latency/HBM-bound random 64/128-byte line reads: every lane owns an
// independent backward-search chain so a wave keeps 64-128 line fetches in flight; no LDS
// staging is used because no two lanes share a block except by accident.
#include "bt2g_kernels.hpp"
#include "bt2g_fm_search.hpp"

// Register budget of the lane-per-task FM kernels.  Measured on MI355X (tools/fm_wpe_sweep.sh, profiles/r02_fm_wpe_sweep.txt):
// asking the compiler for more waves per SIMD (fewer registers) makes every one of them slower -- the spills cost more than
// the extra lanes in flight hide -- except that the 1-mismatch scan is fastest when it is pinned to 2 waves per SIMD.
#ifndef BT2G_MM1_WPE
#define BT2G_MM1_WPE 2
#endif
#define BT2G_FM_BOUNDS __launch_bounds__(256)
#define BT2G_MM1_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BT2G_MM1_WPE, BT2G_MM1_WPE)))

namespace bt2g {

// ------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int comp_base(int c) { return c < 4 ? 3 - c : 4; }

// character p (0-based, Watson orientation) of the fw read or of its reverse complement
__device__ __forceinline__ int read_char(const uint8_t* seq, uint32_t len, uint32_t p, bool rc) {
	return rc ? comp_base(seq[len - 1 - p]) : seq[p];
}

// the copy of the counters this workgroup adds to (DevCounters, bt2g_device.hpp)
__device__ __forceinline__ DevCounters* cnt_slot(DevCounters* cnt) { return cnt + (blockIdx.x & (kCntSlots - 1)); }
__device__ __forceinline__ void wave_add_counter(unsigned long long* ctr, unsigned long long v) {
	// sum over the wave with DPP-free shuffles, one atomic per wave
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
	if ((threadIdx.x & 63) == 0 && v) atomicAdd(ctr, v);
}

// The read as the FM search functions see it: characters through a 16-byte window held in registers.  The searches walk the read
// position by position, so one (unaligned) 16-byte load serves sixteen steps -- a wave's byte load touches 64 different cache lines (one
// per lane) and costs the L1 as much as its 64-byte rank block reads do; rounds 1-3 issued one per step.  The window never reaches past
// the read (the last read of a batch ends where the buffer ends); reads shorter than a window are read byte by byte.
struct GlobRd {
	const uint8_t* s; const uint8_t* q;
	uint32_t len;
	mutable uint32_t base;
	mutable uint4 w;
	__device__ __forceinline__ void init(const uint8_t* s_, const uint8_t* q_, uint32_t len_) { s = s_; q = q_; len = len_; base = 0x80000000u; w = make_uint4(0, 0, 0, 0); }      // (base: no index is within 16 of it)
	__device__ __forceinline__ int seq(uint32_t i) const {
		if (len < 16u) return s[i];
		uint32_t k = i - base;
		if (k >= 16u) {
			base = i & ~15u;
			if (base > len - 16u) base = len - 16u;
			__builtin_memcpy(&w, s + base, 16);
			k = i - base;
		}
		const uint32_t v = (k & 8u) ? ((k & 4u) ? w.w : w.z) : ((k & 4u) ? w.y : w.x);
		return (int)((v >> ((k & 3u) * 8u)) & 0xffu);
	}
	__device__ __forceinline__ int qual(uint32_t i) const { return q[i]; }
	// the sixteen characters lo .. lo + 15 as stored, one per byte (0 past the end of the read)
	__device__ __forceinline__ void window16(uint32_t lo, uint32_t (&o)[4]) const {
		if (len >= 16u) {
			// one 16-byte load in every case: a window that would reach past the read is loaded from the read's last 16 bytes and shifted down
			const uint32_t b = lo + 16u <= len ? lo : len - 16u;
			uint4 v; __builtin_memcpy(&v, s + b, 16);
			uint64_t l64 = (uint64_t)v.x | ((uint64_t)v.y << 32), h64 = (uint64_t)v.z | ((uint64_t)v.w << 32);
			const uint32_t sh = lo - b;                    // 0..15 bytes (lo < len), >= 16: nothing left
			if (sh >= 16u) { l64 = h64 = 0; }
			else if (sh >= 8u) { l64 = h64 >> (8u * (sh - 8u)); h64 = 0; }
			else if (sh > 0u) { l64 = (l64 >> (8u * sh)) | (h64 << (64u - 8u * sh)); h64 >>= 8u * sh; }
			o[0] = (uint32_t)l64; o[1] = (uint32_t)(l64 >> 32); o[2] = (uint32_t)h64; o[3] = (uint32_t)(h64 >> 32);
			return;
		}
		o[0] = o[1] = o[2] = o[3] = 0;
		for (uint32_t k = 0; k < 16u && lo + k < len; k++) o[k >> 2] |= (uint32_t)s[lo + k] << ((k & 3u) * 8u);
	}
	// # of Ns (codes > 3) in the read, sixteen characters per load
	__device__ __forceinline__ uint32_t count_n() const {
		uint32_t ns = 0, i = 0;
		for (; i + 16u <= len; i += 16u) {
			uint4 v; __builtin_memcpy(&v, s + i, 16);
			// a code is 0..4: bit 2 set <=> N
			ns += (uint32_t)__popc(v.x & 0x04040404u) + (uint32_t)__popc(v.y & 0x04040404u) + (uint32_t)__popc(v.z & 0x04040404u) + (uint32_t)__popc(v.w & 0x04040404u);
		}
		for (; i < len; i++) if (s[i] > 3) ns++;
		return ns;
	}
};

// key of the ftabChars-mer seq[off .. off+fc) read in text order (forward index) or reversed
// (mirror index): Ebwt::ftabSeqToInt (bt2_idx.h:1374).  Returns false if an N is present.
template <typename GetC>
__device__ __forceinline__ bool ftab_key(GetC getc, uint32_t off, uint32_t fc, bool text_order, uint64_t& key) {
	key = 0;
	for (uint32_t i = 0; i < fc; i++) {
		const int c = text_order ? getc(off + i) : getc(off + fc - 1 - i);
		if (c > 3) return false;
		key = (key << 2) | (uint64_t)c;
	}
	return true;
}

// ------------------------------------------------------------------------------------
// exact end-to-end sweep
// ------------------------------------------------------------------------------------
template <typename TOff>
__global__ void BT2G_FM_BOUNDS
k_exact_sweep(DevIndex<TOff> ix, bt2g_reads rd, int nofw, int norc, uint32_t mine_max,
              bt2g_sweep_out* __restrict__ out, DevCounters* cnt) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t r = (uint32_t)(gid >> 1);
	const int fwi = (int)(gid & 1);
	unsigned long long nrank = 0, nftab = 0, bwops = 0;
	if (r < rd.n_reads) {
		const DevEbwt<TOff>& e = ix.fw;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		const uint8_t* seq = rd.d_seq + o0;
		const bool rc = fwi == 1;
		const bool skip = (fwi == 0 && nofw) || (fwi == 1 && norc) || len == 0;
		uint32_t mine = 0;
		uint8_t hit = 0;
		TOff top = 0, bot = 0;
		if (!skip) {
			const uint32_t ftab_len = e.ftab_chars;
			uint32_t dep = 0, nedit = 0;
			bool done = false, do_init = true;
			GlobRd g; g.init(seq, nullptr, len);
			auto getc = [&](uint32_t p) -> int { return rc ? comp_base(g.seq(len - 1 - p)) : g.seq(p); };
			while (dep < len && !done) {
				if (do_init) {
					// exactSweepInit (aligner_seed.cpp:752-791)
					top = bot = 0;
					const uint32_t left = len - dep;
					uint64_t key = 0;
					const bool do_ftab = ftab_len > 1 && left >= ftab_len && ftab_key(getc, left - ftab_len, ftab_len, true, key);
					if (do_ftab) {
						top = ftab_hi(e, key);
						bot = ftab_lo(e, key + 1);
						nftab++;
						dep += ftab_len;
					} else {
						const int c = getc(len - dep - 1);
						if (c < 4) { top = e.fchr[c]; bot = e.fchr[c + 1]; }
						dep++;
					}
					if (bot <= top) {           // exactSweepStep (:826-848)
						nedit++;
						if (nedit >= mine_max) { mine = nedit; done = true; }
						continue;
					}
					do_init = false;
				}
				if (dep < len) {
					// exactSweepMapLF (:793-824)
					const int c = getc(len - dep - 1);
					if (c > 3) {
						top = bot = 0;
					} else if (bot - top > 1) {
						bwops += 2;
						TOff nt, nb;
						nrank += rank1_pair(e, top, bot, c, nt, nb);
						top = nt; bot = nb;
					} else {
						bwops++; nrank++;
						const TOff t = map_lf1c(e, top, c);
						if (t == (TOff)OffTraits<TOff>::kMask) { top = bot = 0; }
						else { top = t; bot = t + 1; }
					}
					if (bot <= top) {
						nedit++;
						if (nedit >= mine_max) { mine = nedit; done = true; }
						do_init = true;
					}
					dep++;
				}
			}
			if (!done) {
				mine = nedit;
				if (nedit == 0 && bot > top) hit = 1;
			}
		}
		bt2g_sweep_out* o = out + r;
		o->top[fwi] = hit ? (uint64_t)top : 0;
		o->bot[fwi] = hit ? (uint64_t)bot : 0;
		o->mine[fwi] = mine;
		o->hit[fwi] = hit;
		if (fwi == 0) { for (int i = 0; i < 6; i++) o->pad[i] = 0; }
	}
	wave_add_counter(&cnt_slot(cnt)->rank_queries, nrank);
	wave_add_counter(&cnt_slot(cnt)->ftab_lookups, nftab);
	wave_add_counter(&cnt_slot(cnt)->bwops, bwops);
}

template <typename TOff>
hipError_t launch_exact_sweep(const DevIndex<TOff>& ix, const bt2g_reads& rd, int nofw, int norc, uint32_t mine_max,
                              bt2g_sweep_out* d_out, DevCounters* d_cnt, hipStream_t st) {
	if (rd.n_reads == 0) return hipSuccess;
	const uint64_t nthreads = (uint64_t)rd.n_reads * 2;
	const uint32_t block = 256;
	const uint32_t grid = (uint32_t)((nthreads + block - 1) / block);
	hipLaunchKernelGGL(k_exact_sweep<TOff>, dim3(grid), dim3(block), 0, st, ix, rd, nofw, norc, mine_max, d_out, d_cnt);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// exact (-N 0) multiseed search
// ------------------------------------------------------------------------------------
template <typename TOff>
__global__ void BT2G_FM_BOUNDS
k_seed_search_exact(DevIndex<TOff> ix, bt2g_reads rd, const uint32_t* __restrict__ d_seedlen,
                    const uint32_t* __restrict__ d_interval, const uint32_t* __restrict__ d_offset,
                    const bt2g_read_params* __restrict__ rparams,
                    uint32_t max_seeds, bt2g_seed_hit* __restrict__ out, DevCounters* cnt,
                    uint32_t roundi, ReseedCtl rc_) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t total = (uint64_t)rd.n_reads * 2 * max_seeds;
	unsigned long long nrank = 0, nftab = 0, bwops = 0;
	if (gid < total) {
		// consecutive lanes take consecutive seeds of one (read, strand): their read bytes are adjacent
		const uint32_t i = (uint32_t)(gid % max_seeds);
		const uint64_t rs = gid / max_seeds;
		const uint32_t r = (uint32_t)(rs >> 1);
		const bool rc = (rs & 1) != 0;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		const uint8_t* seq = rd.d_seq + o0;
		uint32_t L = rparams ? (uint32_t)rparams[r].seedlen : d_seedlen[r];
		const uint32_t per = rparams ? (uint32_t)rparams[r].interval : d_interval[r];
		uint32_t off = rparams ? 0u : d_offset[r];
		bool want = true;
		if (roundi > 0) {
			// a re-seeding round: its offset, and whether the previous round was repetitive enough to ask for it
			want = rparams != nullptr && per > 0 && reseed_offset(roundi, seed_rounds_of(rc_.n_seed_rounds, rc_.paired, rparams, r), per, (uint32_t)rparams[r].seedlen, len, off);
			if (want) {
				uint64_t elts = 0; uint32_t nonz = 0;
				for (uint32_t s_ = 0; s_ < 2; s_++) {
					if (s_ == 0 ? rc_.nofw : rc_.norc) continue;
					const bt2g_seed_hit* ph = rc_.prev + ((uint64_t)r * 2 + s_) * max_seeds;
					for (uint32_t k = 0; k < max_seeds; k++) if (ph[k].botf > ph[k].topf) { nonz++; elts += ph[k].botf - ph[k].topf; }
				}
				want = nonz > 0 && elts / nonz >= (uint64_t)rc_.boost_thresh;
			}
		}
		if (L > len) L = len;   // Seed::instantiate shrinks the seed to the read (aligner_seed.cpp:226-230)
		// instantiateSeeds (:523-526)
		uint32_t nseeds = 0;
		if (want && len > 0 && L > 0 && per > 0 && !(off > 0 && (uint64_t)L + off > len)) {
			nseeds = 1;
			if ((int64_t)len - (int64_t)off > (int64_t)L) nseeds += (len - off - L) / per;
		}
		TOff topf = 0, botf = 0, topb = 0, botb = 0;
		bool ok = i < nseeds;
		const uint32_t depth = i * per + off;          // seed's offset from the 5' end
		if (ok && depth + L > len) ok = false;
		if (ok) {
			// Seed sequence as it aligns to the Watson strand: fw -> read[depth, depth+L);
			// rc -> revcomp of that window (instantiateSeq :463-485), i.e. k-th char = comp(read[depth+L-1-k]).
			GlobRd g; g.init(seq, nullptr, len);
			auto getc = [&](uint32_t k) -> int {
				return rc ? comp_base(g.seq(depth + L - 1 - k)) : g.seq(depth + k);
			};
			// An N anywhere disqualifies an exact seed (Constraint::exact cannot absorb it, :338-347)
			for (uint32_t k = 0; k < L; k++) if (getc(k) > 3) { ok = false; break; }
			uint32_t step = 0;
			const uint32_t fc = ix.fw.ftab_chars;
			if (ok) {
				if (fc > 1 && fc <= L) {
					// startSearchSeedBi: ftab jump over the right-most fc characters (:1672-1692)
					uint64_t kf = 0, kb = 0;
					ftab_key(getc, L - fc, fc, true, kf);
					ftab_key(getc, L - fc, fc, false, kb);
					topf = ftab_hi(ix.fw, kf);
					botf = ftab_lo(ix.fw, kf + 1);
					nftab += 2;
					if (botf <= topf) { ok = false; }
					else { topb = ftab_hi(ix.bw, kb); botb = topb + (botf - topf); }
					step = fc;
				} else {
					const int c = getc(L - 1);
					topf = topb = ix.fw.fchr[c];
					botf = botb = ix.fw.fchr[c + 1];
					if (botf <= topf) ok = false;
					step = 1;
				}
			}
			// searchSeedBi main loop for steps[k] = -(L-k): right-to-left over the forward index
			for (; ok && step < L; step++) {
				const int c = getc(L - step - 1);
				if (botf - topf > 1) {
					TOff t[4], b[4];
					bwops++;
					nrank += rank4_pair(ix.fw, topf, botf, t, b);        // mapBiLFEx (bt2_idx.h:2372)
					TOff tp = topb;
					for (int j = 0; j < c; j++) tp += b[j] - t[j];
					if (b[c] == t[c]) { ok = false; break; }
					topf = t[c]; botf = b[c];
					topb = tp; botb = tp + (b[c] - t[c]);
				} else {
					bwops++; nrank++;
					const TOff t = map_lf1c(ix.fw, topf, c);                 // :2003-2016
					if (t == (TOff)OffTraits<TOff>::kMask) { ok = false; break; }
					topf = t; botf = t + 1;
				}
			}
		}
		bt2g_seed_hit h;
		if (ok) { h.topf = topf; h.botf = botf; h.topb = topb; h.botb = botb; }
		else    { h.topf = h.botf = h.topb = h.botb = 0; }
		if (!want) h.topf = ~0ull;       // "this round was not searched for this read": the worker searches it itself should it get there
		out[gid] = h;
	}
	wave_add_counter(&cnt_slot(cnt)->rank_queries, nrank);
	wave_add_counter(&cnt_slot(cnt)->ftab_lookups, nftab);
	wave_add_counter(&cnt_slot(cnt)->bwops, bwops);
}

template <typename TOff>
hipError_t launch_seed_search_exact(const DevIndex<TOff>& ix, const bt2g_reads& rd, const uint32_t* d_seedlen,
                                    const uint32_t* d_interval, const uint32_t* d_offset, const bt2g_read_params* d_rparams,
                                    uint32_t max_seeds, bt2g_seed_hit* d_out, DevCounters* d_cnt, hipStream_t st,
                                    uint32_t roundi, const ReseedCtl* rc) {
	const uint64_t total = (uint64_t)rd.n_reads * 2 * max_seeds;
	if (total == 0) return hipSuccess;
	ReseedCtl ctl; ctl.prev = nullptr; ctl.n_seed_rounds = 0; ctl.boost_thresh = 0; ctl.nofw = ctl.norc = 0; ctl.paired = 0;
	if (rc) ctl = *rc;
	const uint32_t block = 256;
	const uint64_t grid = (total + block - 1) / block;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_seed_search_exact<TOff>, dim3((uint32_t)grid), dim3(block), 0, st, ix, rd, d_seedlen, d_interval,
	                   d_offset, d_rparams, max_seeds, d_out, d_cnt, roundi, ctl);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// SA row -> (reference, offset)
// ------------------------------------------------------------------------------------
template <typename TOff>
__global__ void __launch_bounds__(256)
k_resolve_offsets(DevIndex<TOff> ix, const uint64_t* __restrict__ d_rows, const uint32_t* __restrict__ d_qlen,
                  uint64_t n, int reject_straddle, bt2g_resolved* __restrict__ out, DevCounters* cnt) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long nrank = 0, nsa = 0;
	if (gid < n && d_rows[gid] > (uint64_t)ix.fw.len) {
		// not a row of this index: reject instead of walking garbage
		bt2g_resolved z; z.joined_off = ~0ull; z.tidx = ~0ull; z.toff = 0; z.tlen = 0; z.straddled = 0; z.steps = 0;
		out[gid] = z;
	} else if (gid < n) {
		uint32_t steps = 0;
		const TOff row = (TOff)d_rows[gid];
		const TOff joff = get_offset(ix.fw, row, steps);
		nrank += steps;
		nsa += (row == ix.fw.zoff) ? 0 : 1;
		TOff tidx, toff, tlen;
		bool straddled;
		joined_to_text_off(ix, (TOff)d_qlen[gid], joff, tidx, toff, tlen, reject_straddle != 0, straddled);
		bt2g_resolved o;
		o.joined_off = joff;
		o.tidx = (tidx == (TOff)OffTraits<TOff>::kMask) ? ~0ull : (uint64_t)tidx;
		o.toff = toff; o.tlen = tlen;
		o.straddled = straddled ? 1u : 0u;
		o.steps = steps;
		out[gid] = o;
	}
	wave_add_counter(&cnt_slot(cnt)->rank_queries, nrank);
	wave_add_counter(&cnt_slot(cnt)->sa_lookups, nsa);
}

template <typename TOff>
hipError_t launch_resolve_offsets(const DevIndex<TOff>& ix, const uint64_t* d_rows, const uint32_t* d_qlen, uint64_t n,
                                  int reject_straddle, bt2g_resolved* d_out, DevCounters* d_cnt, hipStream_t st) {
	if (n == 0) return hipSuccess;
	const uint32_t block = 256;
	const uint64_t grid = (n + block - 1) / block;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_resolve_offsets<TOff>, dim3((uint32_t)grid), dim3(block), 0, st, ix, d_rows, d_qlen, n,
	                   reject_straddle, d_out, d_cnt);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// seed-hit extension (SwDriver::extend) for every non-empty round-0 seed hit, one lane per hit
// ------------------------------------------------------------------------------------
// fm_extend_rows_text for ONE row whose compare region lies well inside the joined text and a read of >= 16 characters: the common case of
// k_extend_hits, stripped of every boundary rule (no position can run out of text), with both sides walked by ONE loop -- a lane moves on
// to its right side as soon as its left side ends, so a wave makes as many trips as its longest lane needs in total (~9), not the longest
// left walk plus the longest right walk (~17).  The FM kernels are bound by their instruction stream (profiles/r03zz_pmc_traffic.json:
// 3.7 k vector + 2.2 k scalar wave instructions per READ for this kernel, i.e. ~19 k per wave): what counts is instructions per trip.
__device__ __forceinline__ void extend_one_row_fast(const DevRef& ref, const GlobRd& rd, uint32_t rdlen, uint64_t p, bool fw, uint32_t off, uint32_t len,
                                                    bool right, uint32_t& nlex, uint32_t& nrex, uint32_t& steps_walked) {
	const uint32_t limL = fw ? off : rdlen - len - off, limR = right ? (fw ? rdlen - len - off : off) : 0u;
	uint32_t cnt[2] = {0, 0};
	bool stopped[2] = {false, false};
	uint32_t side = limL > 0 ? 0u : 1u, i0 = 0;
	bool active = (side == 0u ? limL : limR) > 0;
	while (active) {
		const bool left = side == 0u;
		const uint32_t lim = left ? limL : limR;
		const bool down = left == fw;
		uint32_t w4[4];
		const uint32_t lo = down ? (off - i0 >= 16u ? off - i0 - 16u : 0u) : off + len + i0;
		rd.window16(lo, w4);
		uint32_t R = pack4_codes(w4[0]) | (pack4_codes(w4[1]) << 8) | (pack4_codes(w4[2]) << 16) | (pack4_codes(w4[3]) << 24);
		uint32_t N2 = 0;
		if ((w4[0] | w4[1] | w4[2] | w4[3]) & 0x04040404u) {      // an N in the window: rare
			uint32_t N = pack4_nflags(w4[0]) | (pack4_nflags(w4[1]) << 4) | (pack4_nflags(w4[2]) << 8) | (pack4_nflags(w4[3]) << 12);
			if (down) { const uint32_t have = off - i0 >= 16u ? 16u : off - i0; N = rev16((N << (16u - have)) & 0xffffu); }
			N2 = spread_groups(N);
		}
		if (down) { const uint32_t have = off - i0 >= 16u ? 16u : off - i0; R = rev_groups(R << (2u * (16u - have))); }
		if (!fw) R = ~R;
		// the text: sixteen characters in walk order
		const uint64_t q = left ? p - i0 - 16u : p + len + i0;
		uint64_t v; __builtin_memcpy(&v, ref.buf + (q >> 2), 8);
		uint32_t T = (uint32_t)(v >> ((q & 3) << 1));
		if (left) T = rev_groups(T);
		const uint32_t mism = (T ^ R) & ~N2;
		const uint32_t gm = (mism | (mism >> 1)) & 0x55555555u;
		uint32_t k = gm ? (uint32_t)__builtin_ctz(gm) >> 1 : 16u;
		bool end_side = false;
		if (i0 + k >= lim) { k = lim - i0; end_side = true; }            // the side's limit
		else if (k < 16u) { end_side = true; stopped[side] = true; }     // a mismatch
		if (cnt[side] + k >= 255u) { cnt[side] = 255u; end_side = true; stopped[side] = false; } else cnt[side] += k;
		if (!end_side) i0 += 16u;
		else if (left && limR > 0) { side = 1u; i0 = 0; }
		else active = false;
	}
	nlex = cnt[0]; nrex = cnt[1];
	steps_walked = cnt[0] + cnt[1] + (stopped[0] ? 1u : 0u) + (stopped[1] ? 1u : 0u);
}

template <typename TOff>
__global__ void __launch_bounds__(64)
k_extend_hits(DevIndex<TOff> ix, bt2g_reads rd, const bt2g_read_params* __restrict__ rparams, uint32_t max_seeds, int right,
              const bt2g_seed_hit* __restrict__ hits, uint32_t* __restrict__ ext, uint64_t* __restrict__ joffs, DevCounters* cnt,
              uint32_t roundi, uint32_t n_seed_rounds, int paired) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t total = (uint64_t)rd.n_reads * 2 * max_seeds;
	FmCount c; c.bwops = 0; c.sides = 0;
	uint32_t sa = 0;
	if (gid < total) {
		const bt2g_seed_hit h = hits[gid];
		uint32_t e = 0;
		uint64_t jo = kJoffNone;
		if (h.botf > h.topf) {
			const uint32_t i = (uint32_t)(gid % max_seeds);
			const uint64_t rs = gid / max_seeds;
			const uint32_t r = (uint32_t)(rs >> 1);
			const bool fw = (rs & 1) == 0;
			const uint64_t o0 = rd.d_off[r];
			const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
			uint32_t L = (uint32_t)rparams[r].seedlen;
			if (L > len) L = len;
			uint32_t roff = 0;
			if (roundi > 0) reseed_offset(roundi, seed_rounds_of(n_seed_rounds, paired, rparams, r), (uint32_t)rparams[r].interval, (uint32_t)rparams[r].seedlen, len, roff);
			const uint32_t rdoff = i * (uint32_t)rparams[r].interval + roff;
			GlobRd g; g.init(rd.d_seq + o0, rd.d_qual + o0, len);
			uint32_t nlex = 0, nrex = 0;
			const uint64_t nrows = (uint64_t)(h.botf - h.topf);
			if (nrows <= kExtRows) {
				// up to kExtRows rows: resolve their text offsets (the worker needs a unique hit's anyway) and extend by comparing the read with
				// the text of every row (fm_extend_rows_text) -- no chain of rank queries
				uint64_t p[kExtRows];
				bool in_text = true;
				uint32_t steps1 = 0;
#pragma unroll
				for (uint32_t k = 0; k < kExtRows; k++) {
					p[k] = 0;
					if (k < nrows) {
						uint32_t steps = 0;
						const TOff joff = get_offset(ix.fw, (TOff)(h.topf + k), steps);
						p[k] = (uint64_t)joff;
						if (k == 0) steps1 = steps;
						if ((uint64_t)joff >= (uint64_t)ix.fw.len) in_text = false;      // the row of the empty suffix: leave it to the walk
					}
				}
				if (nrows == 1) { c.sides += steps1; c.bwops += steps1; sa++; jo = joff_pack(p[0], steps1); }
				if (in_text) {
					uint32_t walked = 0;
					if (nrows == 1 && len >= 16u && p[0] >= 272u && p[0] + 600u < (uint64_t)ix.fw.len)      // (255 characters either way + the loads' slack)
						extend_one_row_fast(ix.ref, g, len, p[0], fw, rdoff, L, right != 0, nlex, nrex, walked);
					else
					fm_extend_rows_text(ix, g, len, p, (uint32_t)nrows, fw, rdoff, L, nlex, nrex, walked, right != 0);
					if (nrows > 1) { c.bwops += walked; c.sides += walked; sa += (uint32_t)nrows; }
				} else fm_extend_hit(ix, g, len, (TOff)h.topf, (TOff)h.botf, (TOff)h.topb, (TOff)h.botb, fw, rdoff, L, nlex, nrex, c, right != 0);
			} else fm_extend_hit(ix, g, len, (TOff)h.topf, (TOff)h.botf, (TOff)h.topb, (TOff)h.botb, fw, rdoff, L, nlex, nrex, c, right != 0);
			e = nlex | (nrex << 16);
		}
		ext[gid] = e;
		joffs[gid] = jo;
	}
	wave_add_counter(&cnt_slot(cnt)->sa_lookups, sa);
	wave_add_counter(&cnt_slot(cnt)->rank_queries, c.sides);
	wave_add_counter(&cnt_slot(cnt)->bwops, c.bwops);
}

template <typename TOff>
hipError_t launch_extend_hits(const DevIndex<TOff>& ix, const bt2g_reads& rd, const bt2g_read_params* d_rparams, uint32_t max_seeds, int right,
                              const bt2g_seed_hit* d_hits, uint32_t* d_ext, uint64_t* d_joff, DevCounters* d_cnt, hipStream_t st,
                              uint32_t roundi, uint32_t n_seed_rounds, int paired) {
	const uint64_t total = (uint64_t)rd.n_reads * 2 * max_seeds;
	if (total == 0) return hipSuccess;
	// one wavefront per workgroup: a workgroup's slot is held until its slowest lane is done, and the multi-row hits that are still walked
	// (more than kExtRows rows) have a long tail
	const uint64_t grid = (total + 63) / 64;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_extend_hits<TOff>, dim3((uint32_t)grid), dim3(64), 0, st, ix, rd, d_rparams, max_seeds, right, d_hits, d_ext, d_joff, d_cnt, roundi, n_seed_rounds, paired);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Every row's view through the HBM layout (tests: bt2g_index_rows against a walk over the on-disk layout, tests/test_rank_index.py)
// ------------------------------------------------------------------------------------
// out[16] per row: row, joined offset (full suffix array), steps the reference's LF walk to the sample would take, rank4 in the forward
// index, mapLF1 (character, next row), rank4 in the mirror index, the rank pair (row, min(row + 37, len)) of character row & 3 and the
// number of reference sides that pair reads -- the columns tests/hostsim prints for BT2G_INDEX_DUMP.
template <typename TOff>
__global__ void __launch_bounds__(256)
k_index_rows(DevIndex<TOff> ix, uint64_t first, uint64_t n, uint64_t* __restrict__ out) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n) return;
	const uint64_t r = first + gid;
	uint64_t* o = out + gid * 16;
	uint32_t steps = 0;
	const TOff jo = get_offset(ix.fw, (TOff)r, steps);
	TOff f4[4], b4[4];
	rank4(ix.fw, (TOff)r, f4); rank4(ix.bw, (TOff)r, b4);
	TOff rr = (TOff)r;
	const int ch = map_lf1(ix.fw, rr);
	TOff t1 = 0, b1 = 0;
	const int ns = rank1_pair(ix.fw, (TOff)r, (TOff)(r + 37 <= (uint64_t)ix.fw.len ? r + 37 : ix.fw.len), (int)(r & 3), t1, b1);
	o[0] = r; o[1] = (uint64_t)jo; o[2] = steps;
	for (int k = 0; k < 4; k++) { o[3 + k] = (uint64_t)f4[k]; o[9 + k] = (uint64_t)b4[k]; }
	o[7] = (uint64_t)(int64_t)ch; o[8] = ch < 0 ? 0 : (uint64_t)rr;
	o[13] = (uint64_t)t1; o[14] = (uint64_t)b1; o[15] = (uint64_t)ns;
}
template <typename TOff>
hipError_t launch_index_rows(const DevIndex<TOff>& ix, uint64_t first, uint64_t n, uint64_t* d_out, hipStream_t st) {
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_index_rows<TOff>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, ix, first, n, d_out);
	return hipGetLastError();
}
template hipError_t launch_index_rows<uint32_t>(const DevIndex<uint32_t>&, uint64_t, uint64_t, uint64_t*, hipStream_t);
template hipError_t launch_index_rows<uint64_t>(const DevIndex<uint64_t>&, uint64_t, uint64_t, uint64_t*, hipStream_t);

// ------------------------------------------------------------------------------------
// 1-mismatch end-to-end search, one lane per (read, strand, index direction)
// ------------------------------------------------------------------------------------
// Four kernels.  (0) k_one_mm_tasks: one lane per (read, strand) decides whether oneMmSearch runs for it at all -- the worker only
// searches a strand whose exact sweep proved <= 1 edit possible, i.e. about one lane in four -- and lists the (read, strand, index
// direction) combinations that do: the scan then runs on a dense list instead of leaving three quarters of every wave idle.
// (1) k_one_mm_scan: one lane per listed combination walks the exact part of oneMmSearch and,
// wherever a mismatching reference character keeps a non-empty range, queues that branch (Mm1Task) instead of following it.
// (2) k_one_mm_cont: one lane per queued branch finishes it (exact match of the rest of the read).  (3) k_one_mm_fin: per list,
// restore discovery order (increasing depth, then reference character) and publish the count.  On repeat-rich reads the
// branches outnumber the scans by two orders of magnitude and have very uneven lengths: as one flat task list they fill
// the machine; nested inside the scan they left most lanes of a wave waiting for the slowest.
template <typename TOff> struct Mm1Task { uint32_t list; uint16_t dep; uint8_t j, pad; TOff top, bot, topp, botp; };      // list = 0xffffffff: an unused slot of a chunk
constexpr uint32_t kQChunk = 64;

// entry of the scan's task list: list id = read * 4 + strand * 2 + index direction, bit 31 = the read holds one N
__global__ void __launch_bounds__(256)
k_one_mm_tasks(bt2g_align_params P, bt2g_reads rd, const bt2g_read_params* __restrict__ rparams, const bt2g_sweep_out* __restrict__ sweep,
               uint32_t* __restrict__ tasks, unsigned int* __restrict__ tcount) {
	const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool want = false;
	uint32_t ns = 0;
	if (gid < (uint64_t)rd.n_reads * 2) {
		const uint32_t r = (uint32_t)(gid >> 1);
		const bool fw = (gid & 1) == 0;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		// the worker only searches a strand whose exact sweep proved <= 1 edit possible (bt2_search.cpp:3704-3706)
		want = (rparams[r].filt & 15u) == 15u && len >= 2 && sweep[r].mine[fw ? 0 : 1] <= 1 && !(fw ? P.nofw : P.norc);
		if (want) {
			GlobRd g; g.init(rd.d_seq + o0, rd.d_qual + o0, len);
			ns = g.count_n();
			want = ns <= 1;
		}
	}
	// one atomic per wave; the two directions of a strand sit next to each other
	const unsigned long long act = __ballot(want);
	if (act) {
		const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)act) - 1;
		unsigned int base = 0;
		if (lane == leader) base = atomicAdd(tcount, 2u * (unsigned int)__popcll(act));
		base = (unsigned int)__shfl((int)base, leader);
		if (want) {
			const unsigned int idx = base + 2u * (unsigned int)__popcll(act & ((1ull << lane) - 1ull));
			const uint32_t list = (uint32_t)gid * 2u;
			tasks[idx] = list | (ns << 31); tasks[idx + 1] = (list + 1u) | (ns << 31);
		}
	}
}

template <typename TOff>
__global__ void BT2G_MM1_BOUNDS
k_one_mm_scan(DevIndex<TOff> ix, bt2g_align_params P, bt2g_reads rd, const bt2g_read_params* __restrict__ rparams,
              const uint32_t* __restrict__ tasks, const unsigned int* __restrict__ tcount, uint32_t cap, Mm1Hit* __restrict__ out, unsigned int* __restrict__ out_cnt,
              Mm1Task<TOff>* __restrict__ queue, unsigned int* __restrict__ qcount, uint32_t qcap, DevCounters* cnt) {
	FmCount c; c.bwops = 0; c.sides = 0;
	// per wave: the chunk of the branch queue it is filling, and "the queue is full" (sticky).  Whichever lane leads a divergent defer() reads
	// and writes them: volatile, so that no value is carried in a register from one leader's call to another's.
	__shared__ volatile uint32_t s_qbase[4], s_qused[4], s_qfull[4];
	if ((threadIdx.x & 63) == 0) { s_qbase[threadIdx.x >> 6] = 0; s_qused[threadIdx.x >> 6] = kQChunk; s_qfull[threadIdx.x >> 6] = 0; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	const unsigned int nt = *tcount;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t ti = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ti < nt; ti += stride) {
		const uint32_t task = tasks[ti];
		const uint32_t gid = task & 0x7fffffffu, ns = task >> 31;
		const uint32_t r = gid >> 2;
		const bool fw = ((gid >> 1) & 1) == 0;
		const bool ebwtfw = (gid & 1) == 0;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		const bt2g_read_params rp = rparams[r];
		GlobRd g; g.init(rd.d_seq + o0, rd.d_qual + o0, len);
		Mm1Hit* dst = out + (uint64_t)gid * cap;
		auto emit = [&](const Mm1Hit& m) { const unsigned int pos = atomicAdd(&out_cnt[gid], 1u); if (pos < cap) dst[pos] = m; };
		auto defer = [&](uint32_t dep, int j, TOff t, TOff b, TOff tp, TOff bp) -> bool {
			// Queue space comes in chunks of kQChunk slots that a WAVE reserves with one atomic and then hands out from LDS: every branch of
			// every lane used to go through one global counter (an atomic with return per group of lanes arriving together, ~2 M of them per
			// launch, serialised by the L2 channel that owns the word).  What a wave leaves unused of a chunk is marked invalid.
			const unsigned long long act = __ballot(1);
			const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)act) - 1;
			const uint32_t wv = threadIdx.x >> 6;
			unsigned int idx0 = 0xffffffffu;
			if (lane == leader) {
				const uint32_t n = (uint32_t)__popcll(act);
				uint32_t used = s_qused[wv], base = s_qbase[wv];
				if (used + n > kQChunk && !s_qfull[wv]) {
					for (uint32_t k = used; k < kQChunk; k++) queue[base + k].list = 0xffffffffu;      // (used == kQChunk before the first chunk)
					base = atomicAdd(qcount, kQChunk); used = 0;
					// queue full: this wave stops asking (the counter must not wrap around into chunks already handed out)
					if ((uint64_t)base + kQChunk > (uint64_t)qcap) { base = 0; used = kQChunk; s_qbase[wv] = 0; s_qused[wv] = kQChunk; s_qfull[wv] = 1; }
					else s_qbase[wv] = base;
				}
				if (used + n <= kQChunk) { s_qused[wv] = used + n; idx0 = base + used; }
			}
			idx0 = (unsigned int)__shfl((int)idx0, leader);
			if (idx0 == 0xffffffffu) return false;          // queue full: this branch is followed right here
			const unsigned int idx = idx0 + (unsigned int)__popcll(act & ((1ull << lane) - 1ull));
			Mm1Task<TOff> tk; tk.list = gid; tk.dep = (uint16_t)dep; tk.j = (uint8_t)j; tk.pad = 0; tk.top = t; tk.bot = b; tk.topp = tp; tk.botp = bp;
			queue[idx] = tk;
			return true;
		};
		fm_one_mm_dir(ix, P, (int64_t)rp.minsc, rp.nceil, g, len, ns, fw, ebwtfw, emit, c, defer);
	}
	// the rest of the wave's last chunk holds no branch
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	{
		const uint32_t wv = threadIdx.x >> 6, used = s_qused[wv], base = s_qbase[wv];
		for (uint32_t k = used + (threadIdx.x & 63); k < kQChunk; k += 64) queue[base + k].list = 0xffffffffu;
	}
	wave_add_counter(&cnt_slot(cnt)->rank_queries, c.sides);
	wave_add_counter(&cnt_slot(cnt)->bwops, c.bwops);
}

template <typename TOff>
__global__ void BT2G_MM1_BOUNDS
k_one_mm_cont(DevIndex<TOff> ix, bt2g_align_params P, bt2g_reads rd, const bt2g_read_params* __restrict__ rparams, uint32_t cap,
              Mm1Hit* __restrict__ out, unsigned int* __restrict__ out_cnt, const Mm1Task<TOff>* __restrict__ queue,
              const unsigned int* __restrict__ qcount, uint32_t qcap, DevCounters* cnt) {
	FmCount c; c.bwops = 0; c.sides = 0;
	const unsigned int qfull = qcap & ~(kQChunk - 1u);      // whole chunks only: a chunk that did not fit was never handed out
	const unsigned int nq = *qcount < qfull ? *qcount : qfull;
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
		const Mm1Task<TOff> tk = queue[i];
		if (tk.list == 0xffffffffu) continue;
		const uint32_t r = tk.list >> 2;
		const bool fw = ((tk.list >> 1) & 1) == 0, ebwtfw = (tk.list & 1) == 0;
		const uint64_t o0 = rd.d_off[r];
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - o0);
		GlobRd g; g.init(rd.d_seq + o0, rd.d_qual + o0, len);
		Mm1Hit* dst = out + (uint64_t)tk.list * cap;
		fm_one_mm_cont(ix, P, (int64_t)rparams[r].minsc, g, len, fw, ebwtfw, (uint32_t)tk.dep, (int)tk.j, tk.top, tk.bot, tk.topp, tk.botp,
			[&](const Mm1Hit& m) { const unsigned int pos = atomicAdd(&out_cnt[tk.list], 1u); if (pos < cap) dst[pos] = m; }, c);
	}
	wave_add_counter(&cnt_slot(cnt)->rank_queries, c.sides);
	wave_add_counter(&cnt_slot(cnt)->bwops, c.bwops);
}

// discovery order of a list = increasing depth from the end the search direction starts at, then reference character; the
// depth is epos or len-1-epos depending on the (strand, direction) combination, so sorting by epos in the right sense restores it
__global__ void __launch_bounds__(256)
k_one_mm_fin(uint32_t n_lists, uint32_t cap, Mm1Hit* __restrict__ out, const unsigned int* __restrict__ out_cnt, uint8_t* __restrict__ out_n) {
	const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= n_lists) return;
	const unsigned int n = out_cnt[gid];
	if (n > cap) { out_n[gid] = 255; return; }      // more hits than the buffer holds -> the worker redoes this read inline
	out_n[gid] = (uint8_t)n;
	const bool fw = ((gid >> 1) & 1) == 0, ebwtfw = (gid & 1) == 0;
	const bool desc = fw == ebwtfw;                  // depth = len-1-epos: deeper = smaller epos
	Mm1Hit* l = out + (uint64_t)gid * cap;
	for (unsigned int a = 1; a < n; a++) {
		const Mm1Hit v = l[a];
		unsigned int k = a;
		while (k > 0) {
			const Mm1Hit o = l[k - 1];
			const bool o_after = o.epos != v.epos ? (desc ? o.epos < v.epos : o.epos > v.epos) : o.echr > v.echr;
			if (!o_after) break;
			l[k] = o; k--;
		}
		l[k] = v;
	}
}

template <typename TOff>
hipError_t launch_one_mm(const DevIndex<TOff>& ix, const bt2g_align_params& P, const bt2g_reads& rd, const bt2g_read_params* d_rparams,
                         const bt2g_sweep_out* d_sweep, uint32_t cap, void* d_out, uint8_t* d_out_n, unsigned int* d_out_cnt,
                         void* d_queue, uint32_t qcap, unsigned int* d_qcount, uint32_t* d_tasks, DevCounters* d_cnt, hipStream_t st) {
	const uint64_t total = (uint64_t)rd.n_reads * 4;
	if (total == 0) return hipSuccess;
	if (total >= 0x80000000ull) return hipErrorInvalidValue;      // (list ids carry a flag in bit 31)
	unsigned int* d_tcount = d_qcount + 1;      // the two counters of a launch sit side by side (bt2g_capi.hip hands out d_next + 12)
	hipError_t e = hipMemsetAsync(d_out_cnt, 0, total * sizeof(unsigned int), st);
	if (e == hipSuccess) e = hipMemsetAsync(d_qcount, 0, 2 * sizeof(unsigned int), st);
	if (e != hipSuccess) return e;
	const uint64_t grid = (total + 255) / 256;
	hipLaunchKernelGGL(k_one_mm_tasks, dim3((uint32_t)((total / 2 + 255) / 256)), dim3(256), 0, st, P, rd, d_rparams, d_sweep, d_tasks, d_tcount);
	hipLaunchKernelGGL(k_one_mm_scan<TOff>, dim3(256 * 8), dim3(256), 0, st, ix, P, rd, d_rparams, (const uint32_t*)d_tasks, (const unsigned int*)d_tcount, cap, (Mm1Hit*)d_out, d_out_cnt,
	                   (Mm1Task<TOff>*)d_queue, d_qcount, qcap, d_cnt);
	hipLaunchKernelGGL(k_one_mm_cont<TOff>, dim3(256 * 32), dim3(256), 0, st, ix, P, rd, d_rparams, cap, (Mm1Hit*)d_out, d_out_cnt,
	                   (const Mm1Task<TOff>*)d_queue, (const unsigned int*)d_qcount, qcap, d_cnt);
	hipLaunchKernelGGL(k_one_mm_fin, dim3((uint32_t)grid), dim3(256), 0, st, (uint32_t)total, cap, (Mm1Hit*)d_out, (const unsigned int*)d_out_cnt, d_out_n);
	return hipGetLastError();
}
uint64_t one_mm_task_bytes(int off_size) { return off_size == 4 ? sizeof(Mm1Task<uint32_t>) : sizeof(Mm1Task<uint64_t>); }

template hipError_t launch_extend_hits<uint32_t>(const DevIndex<uint32_t>&, const bt2g_reads&, const bt2g_read_params*, uint32_t, int, const bt2g_seed_hit*, uint32_t*, uint64_t*, DevCounters*, hipStream_t, uint32_t, uint32_t, int);
template hipError_t launch_extend_hits<uint64_t>(const DevIndex<uint64_t>&, const bt2g_reads&, const bt2g_read_params*, uint32_t, int, const bt2g_seed_hit*, uint32_t*, uint64_t*, DevCounters*, hipStream_t, uint32_t, uint32_t, int);
template hipError_t launch_one_mm<uint32_t>(const DevIndex<uint32_t>&, const bt2g_align_params&, const bt2g_reads&, const bt2g_read_params*, const bt2g_sweep_out*, uint32_t, void*, uint8_t*, unsigned int*, void*, uint32_t, unsigned int*, uint32_t*, DevCounters*, hipStream_t);
template hipError_t launch_one_mm<uint64_t>(const DevIndex<uint64_t>&, const bt2g_align_params&, const bt2g_reads&, const bt2g_read_params*, const bt2g_sweep_out*, uint32_t, void*, uint8_t*, unsigned int*, void*, uint32_t, unsigned int*, uint32_t*, DevCounters*, hipStream_t);

// ------------------------------------------------------------------------------------
// max over the batch of the number of round-0 seeds per strand (sizes the pre-computation buffers)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_max_seeds(bt2g_reads rd, const bt2g_read_params* __restrict__ rparams, unsigned int* __restrict__ out) {
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned int n = 0;
	if (r < rd.n_reads) {
		const uint32_t len = (uint32_t)(rd.d_off[r + 1] - rd.d_off[r]);
		const uint32_t L = (uint32_t)rparams[r].seedlen, per = (uint32_t)rparams[r].interval;
		n = 1;
		if (len > L && per > 0) n += (len - L) / per;
	}
	for (int o = 32; o > 0; o >>= 1) { const unsigned int v = __shfl_down(n, o); n = v > n ? v : n; }
	if ((threadIdx.x & 63) == 0 && n) atomicMax(out, n);
}

hipError_t launch_max_seeds(const bt2g_reads& rd, const bt2g_read_params* d_rparams, unsigned int* d_out, hipStream_t st) {
	hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned int), st);
	if (e != hipSuccess || rd.n_reads == 0) return e;
	hipLaunchKernelGGL(k_max_seeds, dim3((rd.n_reads + 255) / 256), dim3(256), 0, st, rd, d_rparams, d_out);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Result packing for the trip to the host: the fixed-stride records (sized for BT2G_MAX_EDITS edits per
// alignment, ~1.3 KB) are cut to the edits actually present (~0.15 KB for a typical read).
//   k_pack_sizes : offs[i+1] = bytes of packed record i (offs[0] = 0)
//   k_pack_scan  : in-place inclusive scan of offs[0..n] (one block; 2 MB of u64 for a 262144-read batch)
//   k_pack_copy  : one wavefront per read copies header + trimmed alignments as 32-bit words
// ------------------------------------------------------------------------------------
static constexpr uint32_t kResHead = (uint32_t)offsetof(bt2g_read_result, alns);
static constexpr uint32_t kAlnHead = (uint32_t)offsetof(bt2g_aln, ned);
static_assert(kResHead % 8 == 0 && sizeof(bt2g_aln) % 8 == 0 && sizeof(bt2g_edit) == 6, "packed record layout");
__device__ __forceinline__ uint32_t packed_aln_bytes(uint32_t nned) { return (kAlnHead + nned * (uint32_t)sizeof(bt2g_edit) + 7u) & ~7u; }
// size of a record's alignment slots: the worker class that wrote the record says (bt2g_read_result::pad2 bits 16-31, in units of 8 bytes)
__device__ __forceinline__ uint32_t slot_bytes_of(const bt2g_read_result* rr) { const uint32_t t = rr->pad2 >> 16; return t ? t * 8u : (uint32_t)sizeof(bt2g_aln); }
__device__ __forceinline__ const bt2g_aln* slot_of(const bt2g_read_result* rr, uint32_t slot, uint32_t k) { return (const bt2g_aln*)((const uint8_t*)rr->alns + (uint64_t)k * slot); }
__device__ __forceinline__ uint32_t slot_edits(uint32_t slot) { return (slot - kAlnHead) / (uint32_t)sizeof(bt2g_edit); }

__global__ void __launch_bounds__(256)
k_pack_sizes(const uint8_t* __restrict__ res, uint64_t stride, uint32_t n, uint32_t khits, uint64_t* __restrict__ offs) {
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r == 0) offs[0] = 0;
	if (r >= n) return;
	const bt2g_read_result* rr = (const bt2g_read_result*)(res + (uint64_t)r * stride);
	uint32_t bytes = kResHead;
	const uint32_t na = rr->aligned ? min(rr->nreport, khits) : 0u;
	const uint32_t slot = slot_bytes_of(rr);
	for (uint32_t k = 0; k < na; k++) bytes += packed_aln_bytes(min((uint32_t)slot_of(rr, slot, k)->nned, slot_edits(slot)));
	offs[r + 1] = bytes;
}

__global__ void __launch_bounds__(1024)
k_pack_scan(uint64_t* __restrict__ offs, uint32_t n1) {
	__shared__ uint64_t part[1024];
	const uint32_t t = threadIdx.x, per = (n1 + 1023) / 1024;
	const uint32_t b = min(n1, t * per), e = min(n1, b + per);
	uint64_t sum = 0;
	for (uint32_t i = b; i < e; i++) sum += offs[i];
	part[t] = sum;
	__syncthreads();
	for (uint32_t d = 1; d < 1024; d <<= 1) {
		const uint64_t v = t >= d ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	uint64_t run = part[t] - sum;     // exclusive prefix of this thread's slice
	for (uint32_t i = b; i < e; i++) { run += offs[i]; offs[i] = run; }
}

__global__ void __launch_bounds__(256)
k_pack_copy(const uint8_t* __restrict__ res, uint64_t stride, uint32_t n, uint32_t khits, const uint64_t* __restrict__ offs, uint8_t* __restrict__ out) {
	const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (r >= n) return;
	const uint8_t* src = res + (uint64_t)r * stride;
	uint8_t* dst = out + offs[r];
	const bt2g_read_result* rr = (const bt2g_read_result*)src;
	const uint32_t na = rr->aligned ? min(rr->nreport, khits) : 0u;
	if (lane < kResHead / 4) ((uint32_t*)dst)[lane] = ((const uint32_t*)src)[lane];
	if (lane == 0 && na != rr->nreport) ((bt2g_read_result*)dst)->nreport = na;
	dst += kResHead;
	const uint32_t slot = slot_bytes_of(rr);
	for (uint32_t k = 0; k < na; k++) {
		const uint32_t* a = (const uint32_t*)slot_of(rr, slot, k);
		const uint32_t words = packed_aln_bytes(min((uint32_t)slot_of(rr, slot, k)->nned, slot_edits(slot))) / 4;
		for (uint32_t w = lane; w < words; w += 64) ((uint32_t*)dst)[w] = a[w];
		dst += words * 4;
	}
}

hipError_t launch_pack_results(const void* d_results, uint64_t stride, uint32_t n, uint32_t khits, void* d_packed, uint64_t* d_offsets, hipStream_t st) {
	if (n == 0) return hipMemsetAsync(d_offsets, 0, sizeof(uint64_t), st);
	hipLaunchKernelGGL(k_pack_sizes, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t*)d_results, stride, n, khits, d_offsets);
	hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, d_offsets, n + 1);
	hipLaunchKernelGGL(k_pack_copy, dim3((n + 3) / 4), dim3(256), 0, st, (const uint8_t*)d_results, stride, n, khits, (const uint64_t*)d_offsets, (uint8_t*)d_packed);
	return hipGetLastError();
}

// explicit instantiations
template hipError_t launch_exact_sweep<uint32_t>(const DevIndex<uint32_t>&, const bt2g_reads&, int, int, uint32_t, bt2g_sweep_out*, DevCounters*, hipStream_t);
template hipError_t launch_exact_sweep<uint64_t>(const DevIndex<uint64_t>&, const bt2g_reads&, int, int, uint32_t, bt2g_sweep_out*, DevCounters*, hipStream_t);
template hipError_t launch_seed_search_exact<uint32_t>(const DevIndex<uint32_t>&, const bt2g_reads&, const uint32_t*, const uint32_t*, const uint32_t*, const bt2g_read_params*, uint32_t, bt2g_seed_hit*, DevCounters*, hipStream_t, uint32_t, const ReseedCtl*);
template hipError_t launch_seed_search_exact<uint64_t>(const DevIndex<uint64_t>&, const bt2g_reads&, const uint32_t*, const uint32_t*, const uint32_t*, const bt2g_read_params*, uint32_t, bt2g_seed_hit*, DevCounters*, hipStream_t, uint32_t, const ReseedCtl*);
template hipError_t launch_resolve_offsets<uint32_t>(const DevIndex<uint32_t>&, const uint64_t*, const uint32_t*, uint64_t, int, bt2g_resolved*, DevCounters*, hipStream_t);
template hipError_t launch_resolve_offsets<uint64_t>(const DevIndex<uint64_t>&, const uint64_t*, const uint32_t*, uint64_t, int, bt2g_resolved*, DevCounters*, hipStream_t);

} // namespace bt2g
