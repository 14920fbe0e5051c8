// bt2g_align_core.hpp -- per-read worker logic (see bt2g_align.hpp for the execution model).
//
// `Plat` supplies the wave-parallel pieces:
//    int64 Plat::dp_fill_ee(P, w, fw, rows, cols, scratch, wide)  -> best last-row score
//    void Plat::zero_u16(ptr, n) / sync()
// Everything else is wave-uniform scalar code.
#ifndef BT2G_ALIGN_CORE_HPP_
#define BT2G_ALIGN_CORE_HPP_

#include "bt2g_align.hpp"
#include "bt2g_fm_search.hpp"

namespace bt2g {

template <typename TOff, typename Plat>
struct Aligner {
#define HOT (Plat::hot())
// The index descriptor, the parameter blocks and the batch pre-computation are objects in LDS that the platform hands out
// by name (never through a member reference: an access through a generic pointer is a FLAT instruction, and the wait for a
// FLAT load also waits for every global-memory store still in flight)
#define PRM (Plat::params())                      // const AlignParams&
#define RPR (Plat::rparams())                     // ReadParams& of the loaded read (paired-end mode swaps the mate's in)
#define IX  (Plat::template index<TOff>())        // const DevIndex<TOff>&
#define PRE (Plat::pre())                         // const PreComp*: batch pre-computation (may be null)
	BT2_HD static uint64_t now() { return PRM.profile ? Plat::clock() : 0ull; }      // phase timers only when asked for (bt2g_align_params::profile)

	// The worker's own state -- RNG, current minimum score, DP scratch pointers, pair inputs ... (struct AlState, bt2g_align.hpp) -- is an
	// object in LDS the platform hands out by name (ST), like HOT and the parameter blocks: as members of this class they were reached
	// through `this`, a generic pointer, i.e. by FLAT loads whose wait also drains every global store in flight.  The per-wave work
	// area in HBM is reached through WK.  The class itself has no data members.
#define ST (Plat::st())
#define WK (Plat::work())
// Lane code in one source for both platforms (round 6): a block that every lane executes for itself.  On the device the block runs once, `l` is
// the lane's own number and a LaneReg is the lane's own register; on the host twin it is a loop over the 64 lanes of LaneReg arrays.  What
// crosses lanes (Plat::ballot, Plat::lane, Plat::gather ...) happens between such blocks.
#define BT2_FOR_LANES(l) for (uint32_t l = Plat::lanes_first(); l < 64u; l += Plat::lanes_step())
#define LV(x) (Plat::lv(x, l))
	static constexpr TOff kOffMask = (TOff)OffTraits<TOff>::kMask;

	BT2_HD Aligner(BT2_G Work& w_, DpScratch dp_, uint32_t ridx_ = 0) {
		ST.wp = &w_; ST.dp = dp_; ST.ridx = ridx_; ST.ext_pre = false; ST.pre_ext_cur = nullptr; ST.pre_joff_cur = nullptr;
		ST.pf_steps = ST.pf_tiles = 0; ST.pf_tile_t = 0;
		ST.m_nofw = PRM.nofw != 0; ST.m_norc = PRM.norc != 0; ST.cands_cur = w_.cands;
		ST.n_emit = 0; ST.emit_vmax = 0; ST.emit_on = 1;
	}

	// a fixed-capacity buffer is full: flag the read (its result is not passed off as the reference's) and remember which site
	// noticed first (result record field pad2: diagnostics only)
	BT2_HD void ovf(uint32_t site) { if (!(HOT.err & ERR_OVERFLOW)) HOT.err |= site << 8; HOT.err |= ERR_OVERFLOW; }

	// exact_sweep() from the batch kernel's output
	BT2_HD uint64_t exact_sweep_pre(uint32_t mine[2]) {
		// (what is read from the batch tables is the same in every lane: Plat::uni says so, or the loads' results count as lane-varying and
		// every decision taken on them becomes vector code under an exec mask)
		const bt2g_sweep_out& s = PRE->sweep[Plat::uni(ST.ridx)];
		uint64_t nelt = 0;
		for (int fwi = 0; fwi < 2; fwi++) {
			mine[fwi] = Plat::uni((uint32_t)s.mine[fwi]);
			EEHit& h = HOT.exact[fwi];
			h.top = h.bot = 0;
			if (Plat::uni((uint32_t)s.hit[fwi])) {
				const uint64_t top = Plat::uni((uint64_t)s.top[fwi]), bot = Plat::uni((uint64_t)s.bot[fwi]);
				h.top = top; h.bot = bot; h.fw = fwi == 0 ? 1 : 0; h.has_edit = 0;
				h.score = (int32_t)((int64_t)HOT.len * PRM.match_bonus);
				nelt += bot - top;
			}
		}
		return nelt;
	}

	// one_mm_search() from the batch kernel's output; false if a list overflowed
	BT2_HD bool one_mm_pre(bool nofw_, bool norc_) {
		const bool nofw = Plat::uni((int)nofw_) != 0, norc = Plat::uni((int)norc_) != 0;
		const uint32_t ridx = Plat::uni(ST.ridx);
		const uint32_t n4 = Plat::uni(*reinterpret_cast<const uint32_t*>(PRE->mm1_n + (uint64_t)ridx * 4));      // the four list lengths, one load
		const uint32_t n[4] = {n4 & 0xffu, (n4 >> 8) & 0xffu, (n4 >> 16) & 0xffu, n4 >> 24};
		if ((!nofw && (n[0] == 255 || n[1] == 255)) || (!norc && (n[2] == 255 || n[3] == 255))) return false;
		HOT.n_mm1 = 0; HOT.mm1_elt = 0;
		const int64_t minsc = Plat::uni(ST.minsc);
		for (int k = 0; k < 4; k++) {
			const bool fw = k < 2;
			if ((fw && nofw) || (!fw && norc)) continue;
			const Mm1Hit* src = PRE->mm1 + ((uint64_t)ridx * 4 + k) * PRE->mm1_cap;
			// the batch kernel searched with the read's original minimum score; the worker's may have been tightened since
			for (uint32_t i = 0; i < n[k]; i++) if ((int64_t)Plat::uni(src[i].score) >= minsc) add_mm1(src[i], fw);
		}
		return true;
	}

	// seed_round(offset, interval, seedlen) from the batch kernels' output (`src_all`: the [n_reads][2][max_seeds] table of this round)
	BT2_HD uint32_t seed_round_pre(const bt2g_seed_hit* src_all, uint32_t offset_, uint32_t interval_, uint32_t seedlen_) {
		const uint32_t offset = Plat::uni(offset_), interval = Plat::uni(interval_), seedlen = Plat::uni(seedlen_);
		const uint32_t len = Plat::uni(HOT.len);
		uint32_t nseeds = 1;
		if ((int64_t)len - (int64_t)offset > (int64_t)seedlen) nseeds += (len - offset - seedlen) / interval;
		if (nseeds > (uint32_t)kMaxOffs) { ovf(1); nseeds = kMaxOffs; }
		HOT.num_offs = nseeds;
		HOT.nonz_tot = HOT.nonz_fw = HOT.nonz_rc = 0; HOT.num_elts = 0;
		HOT.n_rank = 0;
		for (uint32_t i = 0; i < nseeds; i++) HOT.off_idx2off[i] = interval * i + offset;
		{
			const bt2g_seed_hit* src = src_all + (uint64_t)ST.ridx * 2 * PRE->max_seeds;
			Plat::load_seed_hits(src, src + PRE->max_seeds, nseeds, ST.m_nofw != 0, ST.m_norc != 0);
		}
		// (pairs: the other mate's row of the same table, for cache_filter's bound on what the round can take from the pool the mates share)
		cache_filter(interval, offset, seedlen, PRM.paired ? src_all + (uint64_t)(ST.ridx ^ 1u) * 2 * PRE->max_seeds : nullptr);
		return 1;   // # instantiated seeds only matters when no seed hit (run() ends the read either way)
	}

	// ---- the reference's per-round seed cache (see struct CacheModel) ----
	BT2_HD void cache_reset() {
		CacheModel& c = HOT.cm;
		const uint64_t bytes = (uint64_t)(PRM.seed_cache_mb > 0 ? PRM.seed_cache_mb : 20) * 1024 * 1024;
		c.pool_total = (uint32_t)((bytes + 16383) / 16384 + 1);       // Pool::Pool (ds.h:3078)
		c.pool_used = 0; c.qn = c.ql = c.san = 0; c.sl = 0; c.nkeys = 0; c.fast_round = 0;
	}
	BT2_HD bool cache_page() { CacheModel& c = HOT.cm; if (c.pool_used == c.pool_total) return false; c.pool_used++; return true; }
	// Seeds of the round in HOT.hits (size = what the search found) -> what SeedResults ends up holding: hits of dropped seeds
	// cleared, esize set, tallies (nonz_*, num_elts) formed.  Exact seeds only: a -N 1 round is tallied unmodelled.
	BT2_HDN void cache_filter(uint32_t interval_, uint32_t offset_, uint32_t seedlen_, const bt2g_seed_hit* other_mate_ = nullptr) {
		const uint32_t interval = Plat::uni(interval_), offset = Plat::uni(offset_), seedlen = Plat::uni(seedlen_);
		CacheModel& c = HOT.cm;
		const uint32_t len = HOT.len, L = seedlen < len ? seedlen : len;
		constexpr uint32_t q_per = sizeof(TOff) == 4 ? 256u : 227u;     // 16 KB / sizeof(RedBlackNode<QKey,QVal>)  (64 / 72 bytes)
		constexpr uint32_t sa_per = sizeof(TOff) == 4 ? 256u : 204u;    // 16 KB / sizeof(RedBlackNode<SAKey,SAVal>) (64 / 80 bytes)
		constexpr uint32_t ql_per = 1024u;                              // 16 KB / sizeof(SAKey)
		constexpr uint64_t sl_per = 16384u / sizeof(TOff);              // 16 KB / sizeof(TIndexOffU)
		HOT.nonz_tot = HOT.nonz_fw = HOT.nonz_rc = 0; HOT.num_elts = 0;
		// Nothing of a round can be dropped or cut when the pool holds all of it even if every seed were a sequence of its own: at most 64 of them, so
		// the three node pools take one page each, and the element list ceil(elements / sl_per).  Then what the search found is what SeedResults ends
		// up holding (esize == size, as the search left it; a sequence that occurs twice has the same range both times), and all that is left to do
		// is the tally -- one lane per seed.  (Only where the pool is this round's and this read's alone: the mates of a pair share one.)
		// Pairs: the mates of a pair share the pool of a round.  The first mate of a round bounds the round by its own ranges plus everything the
		// batch kernels found for the other mate in this round (the other mate's row of the same table: what it will load, or nothing, if its turn
		// comes at all); the flag then covers the second mate's call.  Unknown (the other mate searches its seeds itself): the model runs.
		const bool paired = PRM.paired != 0;
		const bool fresh = Plat::uni(c.nkeys) == 0u && Plat::uni(c.pool_used) == 0u;
		bool covered = paired && Plat::uni(c.fast_round) != 0u;      // the first mate's call of this round bounded both
		bool bound_ok = !paired && fresh;
		uint64_t tot_other = 0;
		// (the other mate must be one the tables hold all seeds of -- else it searches them itself, and more of them than its row shows)
		bool other_fits = false;
		if (paired && PRE) {
			const uint32_t om = (Plat::uni(ST.ridx) & 1u) ^ 1u;
			const uint32_t len2 = Plat::uni(ST.pe_len[om]), iv2 = Plat::uni((uint32_t)ST.pe_rp[om].interval);
			uint32_t L2 = Plat::uni((uint32_t)ST.pe_rp[om].seedlen); if (L2 > len2) L2 = len2;
			other_fits = iv2 > 0u && 1u + (len2 > L2 ? (len2 - L2) / iv2 : 0u) <= Plat::uni(PRE->max_seeds);
		}
		if (paired && !covered && fresh && other_fits && other_mate_ != nullptr && Plat::uni(PRE->max_seeds) <= 32u) {
			const bt2g_seed_hit* const other_mate = Plat::uni_ptr(other_mate_);
			const uint32_t ms = Plat::uni(PRE->max_seeds);
			typename Plat::LaneReg osz, unk;
			BT2_FOR_LANES(l) {
				const uint32_t fwi = l >> 5, i = l & 31u;
				uint32_t v = 0, u = 0;
				if (i < ms) { const bt2g_seed_hit h = other_mate[(uint64_t)fwi * ms + i]; if (h.topf == ~0ull) u = 1u; else if (h.botf > h.topf) v = (uint32_t)(h.botf - h.topf); }
				LV(osz) = v; LV(unk) = u;
			}
			if (Plat::ballot(unk) == 0ull) { tot_other = Plat::lanes_sum(osz); bound_ok = true; }
		}
		if ((covered || bound_ok) && Plat::uni(HOT.num_offs) <= 32u && L <= 32u) {
			const uint32_t n = Plat::uni(HOT.num_offs);
			const bool skf = ST.m_nofw != 0, skr = ST.m_norc != 0;
			typename Plat::LaneReg sz;
			BT2_FOR_LANES(l) { const uint32_t fwi = l >> 5, i = l & 31u; LV(sz) = (i < n && !(fwi ? skr : skf)) ? HOT.hits[fwi][i].size : 0u; }
			const uint64_t tot = Plat::lanes_sum(sz);
			if (covered || 6u + (tot + tot_other + sl_per - 1) / sl_per <= (uint64_t)Plat::uni(c.pool_total)) {
				if (paired) c.fast_round = 1;
				const uint64_t nz = Plat::ballot(sz);
				HOT.nonz_fw = (uint32_t)__builtin_popcountll(nz & 0xffffffffull); HOT.nonz_rc = (uint32_t)__builtin_popcountll(nz >> 32);
				HOT.nonz_tot = HOT.nonz_fw + HOT.nonz_rc; HOT.num_elts = tot;
				return;
			}
		}
		// The key table lives in the arena; its first 64 entries (all of them for an unpaired 150-bp read) are mirrored in lane registers
		// for the duration of this call -- key, length | flags << 8, elements held -- so that a seed costs no memory round trip (it used
		// to cost two: the search of the table and the entry's flags).  Every change is also stored, the arena copy stays complete.
		typename Plat::LaneReg klo, khi, klf, kef;
		Plat::lanes_load_keys(WK.ck_key, WK.ck_len, WK.ck_flags, WK.ck_eff, c.nkeys < 64u ? c.nkeys : 64u, klo, khi, klf, kef);
		auto find = [&](uint64_t key, uint8_t len) -> uint32_t {
			const uint32_t n64 = c.nkeys < 64u ? c.nkeys : 64u;
			const uint32_t e = Plat::find_key_lanes(klo, khi, klf, n64, key, len);
			if (e < n64 || c.nkeys <= 64u) return e < n64 ? e : c.nkeys;
			return 64u + Plat::find_key(WK.ck_key + 64, WK.ck_len + 64, c.nkeys - 64u, key, len);
		};
		auto flags_of = [&](uint32_t e) -> uint32_t { return e < 64u ? (Plat::lane(klf, e) >> 8) & 0xffu : (uint32_t)WK.ck_flags[e]; };
		auto eff_of = [&](uint32_t e) -> uint32_t { return e < 64u ? Plat::lane(kef, e) : WK.ck_eff[e]; };
		auto set_flags = [&](uint32_t e, uint32_t f, uint8_t len) { WK.ck_flags[e] = (uint8_t)f; if (e < 64u) Plat::set_lane(klf, e, (uint32_t)len | (f << 8)); };
		auto set_eff = [&](uint32_t e, uint32_t v) { WK.ck_eff[e] = v; if (e < 64u) Plat::set_lane(kef, e, v); };
		for (int fwi = 0; fwi < 2; fwi++) {
			const bool fw = fwi == 0;
			if ((fw && ST.m_nofw) || (!fw && ST.m_norc)) continue;
			for (uint32_t i = 0; i < HOT.num_offs; i++) {
				HotHit& h = HOT.hits[fwi][i];
				const uint32_t depth = i * interval + offset;
				// the seed as it aligns to the Watson strand, packed; a seed with an N is never instantiated (no cache traffic)
				bool inst = L <= 32;      // (-L > 32 cannot occur)
				const uint64_t key = inst ? Plat::seed_key(fw, depth, L, inst) : 0;
				if (!inst) { h.size = h.esize = 0; continue; }
				const uint32_t e = find(key, (uint8_t)L);
				bool drop = false;
				// beginAlign: the seed sequence enters the QKey map
				if (e == c.nkeys || !(flags_of(e) & 1)) {
					if (c.qn % q_per == 0 && !cache_page()) drop = true;
					else {
						c.qn++;
						if (e == c.nkeys) {
							if (c.nkeys >= (uint32_t)kCacheKeys) { ovf(30); h.size = h.esize = 0; continue; }
							WK.ck_key[e] = key; WK.ck_len[e] = (uint8_t)L;
							if (e < 64u) { Plat::set_lane(klo, e, (uint32_t)key); Plat::set_lane(khi, e, (uint32_t)(key >> 32)); }
							set_flags(e, 0, (uint8_t)L); set_eff(e, 0); c.nkeys++;
						}
						set_flags(e, flags_of(e) | 1u, (uint8_t)L);
					}
				}
				if (!drop && h.size > 0) {
					// addOnTheFly: SAKey list entry, SAKey map node, one element-list slot per row
					if (c.ql % ql_per == 0 && !cache_page()) drop = true;
					else {
						c.ql++;
						if (!(flags_of(e) & 2)) {
							if (c.san % sa_per == 0 && !cache_page()) drop = true;
							else {
								c.san++;
								set_flags(e, flags_of(e) | 2u, (uint8_t)L);
								const uint64_t full = h.size;
								const uint64_t room = (sl_per - c.sl % sl_per) % sl_per + (uint64_t)(c.pool_total - c.pool_used) * sl_per;
								if (full <= room) {
									const uint64_t in_page = (sl_per - c.sl % sl_per) % sl_per;
									if (full > in_page) c.pool_used += (uint32_t)((full - in_page + sl_per - 1) / sl_per);
									c.sl += full; set_eff(e, (uint32_t)full);
								} else {
									c.sl += room; c.pool_used = c.pool_total; set_eff(e, (uint32_t)room);      // the range is cut, this seed is dropped
									drop = true;
								}
							}
						}
					}
				}
				if (drop || h.size == 0) { h.size = h.esize = 0; continue; }
				h.esize = eff_of(e);
				// (esize 0: the pool was already empty when the sequence's range was first stored.  The seed still counts and is
				// ranked, but AlignmentCache::queryQvalImpl hands out no range for it, aligner_cache.h:646)
				HOT.nonz_tot++;
				if (fw) HOT.nonz_fw++; else HOT.nonz_rc++;
				HOT.num_elts += (uint64_t)h.size;
			}
		}
	}

	// =================================================================================
	// A. end-to-end exact / 1-mismatch search and exact seeds
	// =================================================================================

	BT2_HD TOff lf1c(const DevEbwt<TOff>& e, TOff row, int c) { HOT.n_sides++; return map_lf1c(e, row, c); }
	BT2_HD int lf1(const DevEbwt<TOff>& e, TOff& row) { if (row != e.zoff) HOT.n_sides++; return map_lf1(e, row); }

	// One (top,bot) LF step as exactSweepMapLF does (aligner_seed.cpp:793-824)
	BT2_HD void pair_lf(const DevEbwt<TOff>& e, int c, TOff& top, TOff& bot, uint32_t& bwops) {
		if (c > 3) { top = bot = 0; return; }
		if (bot - top > 1) {
			bwops += 2;
			TOff nt, nb;
			HOT.n_sides += rank1_pair(e, top, bot, c, nt, nb);
			top = nt; bot = nb;
		} else {
			bwops++;
			const TOff t = lf1c(e, top, c);
			if (t == kOffMask) { top = bot = 0; } else { top = t; bot = t + 1; }
		}
	}

	// SeedAligner::exactSweep (aligner_seed.cpp:856-970); returns nelt
	BT2_HDN uint64_t exact_sweep(uint32_t mine_max, uint32_t mine[2]) {
		const DevEbwt<TOff>& e = IX.fw;
		const uint32_t len = HOT.len, ftab_len = e.ftab_chars;
		uint64_t nelt = 0;
		HOT.exact[0].top = HOT.exact[0].bot = 0;
		HOT.exact[1].top = HOT.exact[1].bot = 0;
		mine[0] = mine[1] = 0;
		for (int fwi = 0; fwi < 2; fwi++) {
			const bool fw = fwi == 0;
			if ((fw && ST.m_nofw) || (!fw && ST.m_norc)) continue;
			uint32_t dep = 0, nedit = 0;
			bool done = false, do_init = true;
			TOff top = 0, bot = 0;
			while (dep < len && !done) {
				if (do_init) {
					top = bot = 0;
					const uint32_t left = len - dep;
					bool do_ftab = ftab_len > 1 && left >= ftab_len;
					uint64_t key = 0;
					if (do_ftab) {
						for (uint32_t i = 0; i < ftab_len; i++) {
							const int c = rd_char(HOT, HOT.len, fw, left - ftab_len + i);
							if (c > 3) { do_ftab = false; break; }
							key = (key << 2) | (uint64_t)c;
						}
					}
					if (do_ftab) {
						top = ftab_hi(e, key);
						bot = ftab_lo(e, key + 1);
						dep += ftab_len;
					} else {
						const int c = rd_char(HOT, HOT.len, fw, len - dep - 1);
						if (c < 4) { top = e.fchr[c]; bot = e.fchr[c + 1]; }
						dep++;
					}
					if (bot <= top) {
						nedit++;
						if (nedit >= mine_max) { mine[fwi] = nedit; done = true; }
						continue;
					}
					do_init = false;
				}
				if (dep < len) {
					pair_lf(e, rd_char(HOT, HOT.len, fw, len - dep - 1), top, bot, HOT.n_bwops_seed);
					if (bot <= top) {
						nedit++;
						if (nedit >= mine_max) { mine[fwi] = nedit; done = true; }
						do_init = true;
					}
					dep++;
				}
			}
			if (!done) {
				mine[fwi] = nedit;
				if (nedit == 0 && bot > top) {
					EEHit& h = HOT.exact[fwi];
					h.top = top; h.bot = bot; h.fw = fw ? 1 : 0; h.has_edit = 0;
					h.score = (int32_t)((int64_t)len * PRM.match_bonus);
					nelt += (uint64_t)(bot - top);
				}
			}
		}
		return nelt;
	}

	// mapBiLFEx (bt2_idx.h:2372): t/b for all chars in `e`, tp/bp prefix sums starting at topp
	BT2_HD void bi_lf(const DevEbwt<TOff>& e, TOff top, TOff bot, TOff topp, TOff t[4], TOff b[4], TOff tp[4], TOff bp[4]) {
		HOT.n_sides += rank4_pair(e, top, bot, t, b);
		tp[0] = topp;
		bp[0] = tp[0] + (b[0] - t[0]);
		tp[1] = bp[0]; bp[1] = tp[1] + (b[1] - t[1]);
		tp[2] = bp[1]; bp[2] = tp[2] + (b[2] - t[2]);
		tp[3] = bp[2]; bp[3] = tp[3] + (b[3] - t[3]);
	}

	// read accessor for the shared FM search functions
	struct HotRd {
		BT2_HD int seq(uint32_t i) const { return Plat::hot().seq[i]; }
		BT2_HD int qual(uint32_t i) const { return Plat::hot().qual[i]; }
		BT2_HD void window16(uint32_t lo, uint32_t (&o)[4]) const {
			o[0] = o[1] = o[2] = o[3] = 0;
			for (uint32_t k = 0; k < 16u && lo + k < Plat::hot().len; k++) o[k >> 2] |= (uint32_t)Plat::hot().seq[lo + k] << ((k & 3u) * 8u);
		}
	};

	// SeedAligner::oneMmSearch with repex=false, rep1mm=true (aligner_seed.cpp:975-1325)
	BT2_HDN void one_mm_search(bool nofw, bool norc) {
		const uint32_t len = HOT.len;
		HOT.n_mm1 = 0; HOT.mm1_elt = 0;
		uint32_t ns = 0;
		for (uint32_t i = 0; i < len; i++) if (HOT.seq[i] > 3) ns++;
		if (ns > 1) return;
		FmCount cnt; cnt.bwops = 0; cnt.sides = 0;
		HotRd rd;
		for (int fwi = 0; fwi < 2; fwi++) {
			const bool fw = fwi == 0;
			if ((fw && nofw) || (!fw && norc)) continue;
			for (int ebwtfwi = 0; ebwtfwi < 2; ebwtfwi++) {
				fm_one_mm_dir(IX, PRM, ST.minsc, RPR.nceil, rd, len, ns, fw, ebwtfwi == 0,   // ST.minsc[mate] as tightened so far (bt2_search.cpp:3712)
					[&](const Mm1Hit& m) { add_mm1(m, fw); }, cnt);
			}
		}
		HOT.n_bwops_seed += cnt.bwops; HOT.n_sides += cnt.sides;
	}

	BT2_HD void add_mm1(const Mm1Hit& m, bool fw) {
		const uint32_t nm = Plat::uni(HOT.n_mm1);
		if (nm >= (uint32_t)kMaxMm1) { ovf(2); return; }
		BT2_G EEHit& h = WK.mm1[nm]; HOT.n_mm1 = nm + 1;
		h.top = m.top; h.bot = m.bot; h.score = m.score;
		h.epos = m.epos; h.echr = m.echr; h.eqchr = m.eqchr;
		h.fw = fw ? 1 : 0; h.has_edit = 1;
		HOT.mm1_elt += (uint64_t)(m.bot - m.top);
	}

	// One -N 0 seeding round: Seed::mmSeeds + instantiateSeeds + searchAllSeeds
	// (aligner_seed.cpp:498-720,1638-2037).  Returns # instantiated seeds.
	BT2_HDN uint32_t seed_round(uint32_t offset_, uint32_t interval_, uint32_t seedlen_) {
		const uint32_t interval = Plat::uni(interval_), offset = Plat::uni(offset_), seedlen = Plat::uni(seedlen_);
		const uint32_t len = HOT.len;
		uint32_t L = seedlen < len ? seedlen : len;
		uint32_t nseeds = 1;
		if ((int64_t)len - (int64_t)offset > (int64_t)seedlen) nseeds += (len - offset - seedlen) / interval;
		if (nseeds > (uint32_t)kMaxOffs) { ovf(3); nseeds = kMaxOffs; }
		HOT.num_offs = nseeds;
		HOT.nonz_tot = HOT.nonz_fw = HOT.nonz_rc = 0; HOT.num_elts = 0;
		HOT.n_rank = 0;
		uint32_t ninst = 0;
		for (uint32_t i = 0; i < nseeds; i++) HOT.off_idx2off[i] = interval * i + offset;
		const uint32_t fc = IX.fw.ftab_chars;
		for (int fwi = 0; fwi < 2; fwi++) {
			const bool fw = fwi == 0;
			for (uint32_t i = 0; i < nseeds; i++) {
				HotHit& h = HOT.hits[fwi][i];
				h.topf = h.topb = 0; h.size = h.esize = 0;
				HOT.sorted[fwi][i] = 0;
			}
			if ((fw && ST.m_nofw) || (!fw && ST.m_norc)) continue;
			for (uint32_t i = 0; i < nseeds; i++) {
				const uint32_t depth = i * interval + offset;
				// seed char k as it aligns to the Watson strand (instantiateSeq :463-485)
				auto getc = [&](uint32_t k) -> int { return fw ? (int)HOT.seq[depth + k] : comp4(HOT.seq[depth + L - 1 - k]); };
				bool ok = true;
				for (uint32_t k = 0; k < L; k++) if (getc(k) > 3) { ok = false; break; }
				if (!ok) continue;      // Seed::instantiate fails: exact zones cannot absorb an N
				ninst++;
				TOff topf = 0, botf = 0, topb = 0, botb = 0;
				uint32_t step = 0;
				if (fc > 1 && fc <= L) {
					uint64_t kf = 0, kb = 0;
					for (uint32_t k = 0; k < fc; k++) {
						kf = (kf << 2) | (uint64_t)getc(L - fc + k);
						kb = (kb << 2) | (uint64_t)getc(L - 1 - k);
					}
					topf = ftab_hi(IX.fw, kf); botf = ftab_lo(IX.fw, kf + 1);
					if (botf <= topf) continue;
					topb = ftab_hi(IX.bw, kb); botb = topb + (botf - topf);
					step = fc;
				} else {
					const int c = getc(L - 1);
					topf = topb = IX.fw.fchr[c];
					botf = botb = IX.fw.fchr[c + 1];
					if (botf <= topf) continue;
					step = 1;
				}
				for (; ok && step < L; step++) {
					const int c = getc(L - step - 1);
					if (botf - topf > 1) {
						TOff t[4], b[4];
						HOT.n_bwops_seed++;
						HOT.n_sides += rank4_pair(IX.fw, topf, botf, t, b);
						TOff tp = topb;
						for (int j = 0; j < c; j++) tp += b[j] - t[j];
						if (b[c] == t[c]) { ok = false; break; }
						topf = t[c]; botf = b[c]; topb = tp; botb = tp + (b[c] - t[c]);
					} else {
						HOT.n_bwops_seed++;
						const TOff t = lf1c(IX.fw, topf, c);
						if (t == kOffMask) { ok = false; break; }
						topf = t; botf = t + 1;
					}
				}
				if (!ok) continue;
				HotHit& h = HOT.hits[fwi][i];
				h.topf = topf; h.topb = topb; h.size = h.esize = (uint32_t)(botf - topf);
			}
		}
		cache_filter(interval, offset, seedlen);      // SeedResults::add (aligner_seed.h:639-676) after the cache has had its say
		return ninst;
	}

	// One -N 1 seeding round: Seed::oneMmSeeds (aligner_seed.cpp:380-399) instantiated at every offset and searched with
	// searchSeedBi (:1858-2037).  Two half-and-half policies per (strand, offset): left-to-right on the mirror index,
	// exact over the left ceil(L/2) characters and up to one mismatch over the rest; right-to-left on the forward
	// index, exact over the right floor(L/2) characters and exactly one mismatch over the rest.  Every hit (one BW
	// range per distinct reference string) is kept; HOT.hits[..] then indexes a run of Work::sranges.
	// An N may only fall in a policy's mismatch zone, where it stands for the one mismatch (Seed::instantiate :325-345).
	// -N 1: what the reference's per-round seed cache does with the hits [first, last) of one seed (see cache_filter for the exact-seed
	// form).  Returns false when the seed is dropped (out of memory in beginAlign or in any addOnTheFly: searchAllSeeds counts an "oom" and
	// skips sr.add, aligner_seed.cpp:672-690); the pool, the maps and the ranges stored so far keep what was spent.  Sets
	// SeedRange::esize = elements the cache holds for the range's reference string (cut when the pool ran out while it was stored;
	// the first stored copy decides for every later seed that hits the same string).
	BT2_HD bool cache_account_mm1(uint32_t first, uint32_t last, uint64_t qkey, bool qcacheable, uint32_t L) {
		CacheModel& c = HOT.cm;
		constexpr uint32_t q_per = sizeof(TOff) == 4 ? 256u : 227u;
		constexpr uint32_t sa_per = sizeof(TOff) == 4 ? 256u : 204u;
		constexpr uint32_t ql_per = 1024u;
		constexpr uint64_t sl_per = 16384u / sizeof(TOff);
		// beginAlign: a seed without N enters the QKey map (one with an N is not cacheable: no node)
		if (qcacheable) {
			const uint32_t e = Plat::find_key(WK.ck_key, WK.ck_len, c.nkeys, qkey, (uint8_t)L);
			if (e == c.nkeys || !(WK.ck_flags[e] & 1)) {
				if (c.qn % q_per == 0 && !cache_page()) return false;
				c.qn++;
				if (e == c.nkeys) {
					if (c.nkeys >= (uint32_t)kCacheKeys) { ovf(30); return false; }
					WK.ck_key[e] = qkey; WK.ck_len[e] = (uint8_t)L; WK.ck_flags[e] = 0; WK.ck_eff[e] = 0; c.nkeys++;
				}
				WK.ck_flags[e] |= 1;
			}
		}
		bool ok = true;
		for (uint32_t r = first; r < last; r++) {
			BT2_G SeedRange& sr = WK.sranges[r];
			// addOnTheFly: SAKey list entry, SAKey map node, one element-list slot per row
			if (c.ql % ql_per == 0 && !cache_page()) { ok = false; sr.esize = 0; continue; }
			c.ql++;
			const uint64_t key = WK.srange_key[r];
			uint32_t e = Plat::find_key(WK.ck_key, WK.ck_len, c.nkeys, key, (uint8_t)L);
			if (e == c.nkeys) {
				if (c.nkeys >= (uint32_t)kCacheKeys) { ovf(30); ok = false; sr.esize = 0; continue; }
				WK.ck_key[e] = key; WK.ck_len[e] = (uint8_t)L; WK.ck_flags[e] = 0; WK.ck_eff[e] = 0; c.nkeys++;
			}
			if (!(WK.ck_flags[e] & 2)) {
				if (c.san % sa_per == 0 && !cache_page()) { ok = false; sr.esize = 0; continue; }
				c.san++;
				WK.ck_flags[e] |= 2;
				const uint64_t full = sr.size;
				const uint64_t room = (sl_per - c.sl % sl_per) % sl_per + (uint64_t)(c.pool_total - c.pool_used) * sl_per;
				if (full <= room) {
					const uint64_t in_page = (sl_per - c.sl % sl_per) % sl_per;
					if (full > in_page) c.pool_used += (uint32_t)((full - in_page + sl_per - 1) / sl_per);
					c.sl += full; WK.ck_eff[e] = (uint32_t)full;
				} else {
					c.sl += room; c.pool_used = c.pool_total; WK.ck_eff[e] = (uint32_t)room;      // the range is cut
					ok = false;
				}
			}
			sr.esize = WK.ck_eff[e];
		}
		return ok;
	}

	BT2_HDN uint32_t seed_round_mm1(uint32_t offset, uint32_t interval, uint32_t seedlen) {
		const uint32_t len = HOT.len;
		const uint32_t L = seedlen < len ? seedlen : len;
		uint32_t nseeds = 1;
		if ((int64_t)len - (int64_t)offset > (int64_t)seedlen) nseeds += (len - offset - seedlen) / interval;
		if (nseeds > (uint32_t)kMaxOffs) { ovf(4); nseeds = kMaxOffs; }
		HOT.num_offs = nseeds;
		HOT.nonz_tot = HOT.nonz_fw = HOT.nonz_rc = 0; HOT.num_elts = 0;
		HOT.n_rank = 0;
		uint32_t ninst = 0, nsr = 0;
		for (uint32_t i = 0; i < nseeds; i++) HOT.off_idx2off[i] = interval * i + offset;
		const uint32_t fc = IX.fw.ftab_chars;
		for (int fwi = 0; fwi < 2; fwi++) {
			const bool fw = fwi == 0;
			for (uint32_t i = 0; i < nseeds; i++) {
				HotHit& h = HOT.hits[fwi][i];
				h.topf = h.topb = 0; h.size = h.esize = 0;
				HOT.sorted[fwi][i] = 0;
			}
			if ((fw && ST.m_nofw) || (!fw && ST.m_norc)) continue;
			for (uint32_t i = 0; i < nseeds; i++) {
				const uint32_t depth = i * interval + offset;
				auto getc = [&](uint32_t k) -> int { return fw ? (int)HOT.seq[depth + k] : comp4(HOT.seq[depth + L - 1 - k]); };
				const uint32_t first = nsr;
				uint64_t elts = 0;
				uint32_t inst_here = 0;
				// the seed as a packed key (an N as 0); the reference string of a hit is the seed with position `sp` replaced by `sc`
				bool qcacheable = true;
				const uint64_t qkey = L <= 32 ? Plat::seed_key(fw, depth, L, qcacheable) : 0ull;
				auto report = [&](TOff topf, TOff botf, TOff topb, uint32_t sp, int sc) {
					if (nsr >= (uint32_t)kMaxSat2) { ovf(5); return; }
					uint64_t key = qkey;
					if (sp < L) { const uint32_t sh = 2u * (L - 1u - sp); key = (key & ~(3ull << sh)) | ((uint64_t)(sc & 3) << sh); }
					WK.srange_key[nsr] = key;
					BT2_G SeedRange& r = WK.sranges[nsr++];
					r.topf = topf; r.topb = topb; r.size = r.esize = (uint32_t)(botf - topf);
					elts += (uint64_t)(botf - topf);
				};
				// The reference advances the searches of a seed's two policies in LOCKSTEP, one step each per turn (searchSeedBi's loop over
				// paramVec, aligner_seed.cpp:1878-2037; a mismatch branch runs to its end at once, :1964), so the hits of a seed are discovered
				// interleaved -- which is the order the seed cache stores them in, and when its pool runs out, the order decides which
				// ranges are cut (cache_account_mm1).
				struct PolState { bool live; uint32_t step; int mms; TOff topf, botf, topb, botb; } ps[2];
				auto pos_of = [&](int pol, uint32_t k) -> uint32_t { return pol == 0 ? k : L - 1 - k; };
				auto zone1_of = [&](int pol, uint32_t k) -> bool { const uint32_t z0 = pol == 0 ? (L + 1) / 2 : L / 2; return k >= z0 || k == L - 1; };
				// startSearchSeedBi of both policies, in order
				for (int pol = 0; pol < 2; pol++) {
					const bool ltr = pol == 0;
					PolState& s = ps[pol];
					s.live = false; s.step = 0; s.mms = 1; s.topf = s.botf = s.topb = s.botb = 0;
					// step k reads seed character pos(k); zone(k) = 1 once past the exact half (Seed::instantiate :256-276)
					auto pos = [&](uint32_t k) -> uint32_t { return pos_of(pol, k); };
					auto zone1 = [&](uint32_t k) -> bool { return zone1_of(pol, k); };       // steps [0, z0) are zone 0, except that the last step always closes zone 1
					int mms = 1;
					bool inst = true;
					for (uint32_t k = 0; k < L && inst; k++) {
						if (getc(pos(k)) > 3) { if (zone1(k) && mms > 0) mms--; else inst = false; }
					}
					if (!inst) continue;
					ninst++; inst_here++;
					s.mms = mms;
					// maxjump: leading steps that stay in zone 0 (for the right-to-left seed the insertion zone changes at the same step)
					uint32_t maxjump = 0;
					while (maxjump < L && !zone1(maxjump)) maxjump++;
					TOff topf = 0, botf = 0, topb = 0, botb = 0;
					uint32_t step = 0;
					if (fc > 1 && fc <= maxjump) {
						const uint32_t off = ltr ? 0 : L - fc;
						uint64_t kf = 0, kb = 0;
						for (uint32_t k = 0; k < fc; k++) {
							kf = (kf << 2) | (uint64_t)getc(off + k);
							kb = (kb << 2) | (uint64_t)getc(off + fc - 1 - k);
						}
						topf = ftab_hi(IX.fw, kf); botf = ftab_lo(IX.fw, kf + 1);
						if (botf <= topf) continue;
						topb = ftab_hi(IX.bw, kb); botb = topb + (botf - topf);
						step = fc;
					} else if (maxjump > 0) {
						const int c = getc(pos(0));
						topf = topb = IX.fw.fchr[c];
						botf = botb = IX.fw.fchr[c + 1];
						if (botf <= topf) continue;
						step = 1;
					} else {
						topf = topb = 0;
						botf = botb = IX.fw.fchr[4];
					}
					if (step == L) { report(topf, botf, topb, L, 0); continue; }
					s.live = true; s.step = step; s.topf = topf; s.botf = botf; s.topb = topb; s.botb = botb;
				}
				// one step of one policy
				auto do_step = [&](int pol) {
					const bool ltr = pol == 0;
					PolState& s = ps[pol];
					auto pos = [&](uint32_t k) -> uint32_t { return pos_of(pol, k); };
					auto zone1 = [&](uint32_t k) -> bool { return zone1_of(pol, k); };
					const int ceil1 = ltr ? 0x7fffffff : 0;               // mmsCeil of zone 1: the right-to-left policy must have spent its mismatch
					const int mms = s.mms;
					const DevEbwt<TOff>& e = ltr ? IX.bw : IX.fw;
					TOff& topf = s.topf; TOff& botf = s.botf; TOff& topb = s.topb; TOff& botb = s.botb;
					// exact continuation after the mismatch (the recursive searchSeedBi call: every zone is used up)
					auto finish_exact = [&](uint32_t st, TOff tf_, TOff bf_, TOff tb_, TOff bb_, uint32_t sp, int sc) {
						for (; st < L; st++) {
							const int c = getc(pos(st));
							TOff& top = ltr ? tb_ : tf_; TOff& bot = ltr ? bb_ : bf_;
							TOff& topp = ltr ? tf_ : tb_; TOff& botp = ltr ? bf_ : bb_;
							HOT.n_bwops_seed++;
							if (bot - top > 1) {
								TOff t[4], b[4], tp[4], bp[4];
								bi_lf(e, top, bot, topp, t, b, tp, bp);
								if (b[c] == t[c]) return;
								top = t[c]; bot = b[c]; topp = tp[c]; botp = bp[c];
							} else {
								const TOff t = lf1c(e, top, c);
								if (t == kOffMask) return;
								top = t; bot = t + 1;
							}
						}
						report(tf_, bf_, tb_, sp, sc);
					};
					const uint32_t k = s.step;
					const int c = getc(pos(k));
					const bool z1 = zone1(k), leave = k == L - 1;
					TOff& top = ltr ? topb : topf; TOff& bot = ltr ? botb : botf;      // range in the index being walked
					TOff& topp = ltr ? topf : topb; TOff& botp = ltr ? botf : botb;    // and in the other one
					TOff t[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, tp[4], bp[4];
					tp[0] = tp[1] = tp[2] = tp[3] = topp; bp[0] = bp[1] = bp[2] = bp[3] = botp;
					const bool wide = bot - top > 1;
					if (wide) { HOT.n_bwops_seed++; bi_lf(e, top, bot, topp, t, b, tp, bp); }
					bool probed = false;     // single-row range whose character was looked up by the edit branch
					TOff row1 = top;
					if ((z1 && mms > 0) || c > 3) {
						bool bail = false;
						if (!wide) {
							HOT.n_bwops_seed++;
							const int cc = lf1(e, row1);
							if (cc < 0) bail = true; else { t[cc] = row1; b[cc] = row1 + 1; probed = true; }
						}
						if (!bail) {
							const int after = c > 3 ? mms : mms - 1;
							if (!leave || after <= ceil1) {
								for (int j = 0; j < 4; j++) {
									if (j == c || b[j] == t[j]) continue;
									const TOff jt = t[j], jb = b[j], jtp = tp[j], jbp = bp[j];
									if (ltr) finish_exact(k + 1, jtp, jbp, jt, jb, pos(k), j); else finish_exact(k + 1, jt, jb, jtp, jbp, pos(k), j);
								}
							}
						}
					}
					if (c > 3) { s.live = false; return; }
					if (leave && z1 && mms > ceil1) { s.live = false; return; }
					if (!wide) {
						HOT.n_bwops_seed++;
						if (probed) {
							// mapLF1(ntop, tloc, c) after mapLF1(ntop&, tloc): the row test sees the already-advanced row (:1995 after :1918)
							if (t[c] == b[c] || row1 == e.zoff) { s.live = false; return; }
						} else {
							const TOff r = lf1c(e, top, c);
							if (r == kOffMask) { s.live = false; return; }
							t[c] = r; b[c] = r + 1;
						}
					}
					if (b[c] == t[c]) { s.live = false; return; }
					top = t[c]; bot = b[c]; topp = tp[c]; botp = bp[c];
					s.step++;
					if (s.step == L) { report(topf, botf, topb, L, 0); s.live = false; }
				};
				while (ps[0].live || ps[1].live) {
					for (int pol = 0; pol < 2; pol++) if (ps[pol].live) do_step(pol);
				}
				// the seed cache's page accounting for this seed (CacheModel): SeedSearchCache::beginAlign, then addOnTheFly per hit in
				// discovery order (aligner_seed.h:1483-1510, aligner_cache.cpp:53-105); a seed whose hits did not all fit is dropped
				if (inst_here > 0 && L <= 32 && !cache_account_mm1(first, nsr, qkey, qcacheable, L)) continue;
				if (nsr > first && elts > 0) {
					HotHit& h = HOT.hits[fwi][i];
					h.topf = first; h.topb = nsr - first; h.size = h.esize = (uint32_t)elts;
					HOT.nonz_tot++;
					if (fw) HOT.nonz_fw++; else HOT.nonz_rc++;
					HOT.num_elts += elts;
				}
			}
		}
		HOT.n_sranges = nsr;
		return ninst;
	}

	BT2_HD uint64_t hit_elts(int fwi, uint32_t i) const { return HOT.hits[fwi][i].size; }

	// SeedResults::rankSeedHits, all=false (aligner_seed.h:1019-1080)
	BT2_HDN void rank_seed_hits() {
		HOT.n_rank = 0;
		if (PRM.all_hits) {
			// rankSeedHits(all = true): no random draws; offsets 1.. first (fw then rc), offset 0 last (aligner_seed.h:1020-1038)
			auto push = [&](uint32_t i, bool fw) {
				if (HOT.n_rank >= (uint32_t)kMaxRanges) { ovf(6); return; }
				HOT.rank_offs[HOT.n_rank] = (uint8_t)i; HOT.rank_fw[HOT.n_rank] = fw ? 1 : 0; HOT.n_rank++;
			};
			for (uint32_t i = 1; i < HOT.num_offs; i++) for (int fwi = 0; fwi < 2; fwi++) if (HOT.hits[fwi][i].size > 0) push(i, fwi == 0);
			if (HOT.num_offs > 0) { if (HOT.hits[0][0].size > 0) push(0, true); if (HOT.hits[1][0].size > 0) push(0, false); }
			return;
		}
		while (HOT.n_rank < HOT.nonz_tot) {
			uint64_t minsz = 0xffffffffull;      // MAX_U32 even for large indexes, as in the reference
			uint32_t minidx = 0;
			bool minfw = true;
			const bool rb = ST.rnd.nextBool();
			for (int fwi = 0; fwi <= 1; fwi++) {
				const bool fw = (fwi == (rb ? 1 : 0));
				const int s = fw ? 0 : 1;
				uint32_t i = ST.rnd.nextU32() % HOT.num_offs;
				for (uint32_t ii = 0; ii < HOT.num_offs; ii++) {
					const uint64_t ne = hit_elts(s, i);
					if (ne > 0 && !HOT.sorted[s][i] && (TOff)ne < (TOff)minsz) {
						minsz = ne; minidx = i; minfw = fw;
					}
					if ((++i) == HOT.num_offs) i = 0;
				}
			}
			HOT.sorted[minfw ? 0 : 1][minidx] = 1;
			HOT.rank_offs[HOT.n_rank] = minidx;
			HOT.rank_fw[HOT.n_rank] = minfw ? 1 : 0;
			HOT.n_rank++;
		}
	}

	// =================================================================================
	// B. Random1toN / RowSampler
	// =================================================================================
	BT2_HD void r1n_init(BT2_G R1N& r, uint32_t n, bool without_replacement) {
		r.sz = r.n = n;
		r.converted = 0;
		r.swaplist = (n < 128 || without_replacement) ? 1 : 0;
		r.cur = 0;
		r.list_off = r.list_len = r.seen_off = r.seen_len = 0;
		uint32_t th = (uint32_t)(0.10f * (float)n);
		r.thresh = th > 16 ? th : 16;
		r.inited = 1;
	}
	BT2_HD void r1n_reset(BT2_G R1N& r) { r.sz = r.n = r.cur = 0; r.swaplist = r.converted = 0; r.list_len = r.seen_len = 0; r.thresh = 0; r.inited = 0; }
	BT2_HD bool r1n_done(const BT2_G R1N& r) const { return r.n > 0 && r.cur >= r.n; }
	BT2_HD void r1n_set_done(BT2_G R1N& r) { r.cur = r.n; }
	BT2_HD void r1n_init_seq(BT2_G R1N& r, uint32_t n) { r1n_reset(r); r.sz = r.n = n; r.swaplist = 2; r.inited = 1; }
	BT2_HD uint32_t lists_alloc(uint32_t n) {
		if (HOT.lists_used + n > (uint32_t)kListArena) { ovf(7); return 0; }
		const uint32_t o = HOT.lists_used;
		HOT.lists_used += n;
		return o;
	}
	BT2_HDN uint32_t r1n_next(BT2_G R1N& r_) {
		BT2_G R1N& r = *Plat::uni_ptr(&r_);      // (a call argument: lane-varying to the compiler unless told otherwise)
		if (r.swaplist == 2) return r.cur++;          // -d: rows in index order (currIdx++, aligner_sw_driver.cpp:1120)
		if (r.cur == 0 && !r.converted) {
			if (r.n == 1) { r.cur = 1; return 0; }
			if (r.swaplist) {
				r.list_off = lists_alloc(r.n);
				r.list_len = r.n;
				for (uint32_t i = 0; i < r.n; i++) WK.lists[r.list_off + i] = i;
			}
		}
		if (r.swaplist) {
			const uint32_t rr = r.cur + (ST.rnd.nextU32() % (r.n - r.cur));
			BT2_G uint32_t* l = WK.lists + r.list_off;
			if (rr != r.cur) { const uint32_t tmp = l[r.cur]; l[r.cur] = l[rr]; l[rr] = tmp; }
			return l[r.cur++];
		}
		// seen-list mode (n >= 128)
		// the seen list can never hold more entries than rows are drawn for one read (max_iters), however large the range:
		// a 100 000-row repeat range has thresh = 10 000 but is asked for a few hundred rows at most
		if (r.seen_len == 0 && r.cur == 0) { const uint32_t cap = (uint32_t)PRM.max_iters + 2; r.seen_off = lists_alloc(r.thresh + 1 < cap ? r.thresh + 1 : cap); }
		BT2_G uint32_t* seen = WK.lists + r.seen_off;
		const uint32_t seen_sz = r.seen_len;
		uint32_t rn = 0;
		bool again = true;
		while (again) {
			rn = ST.rnd.nextU32() % r.n;
			again = false;
			for (uint32_t i = 0; i < seen_sz; i++) if (seen[i] == rn) { again = true; break; }
		}
		{ const uint32_t cap = (uint32_t)PRM.max_iters + 2; if (r.seen_len >= (r.thresh + 1 < cap ? r.thresh + 1 : cap)) { ovf(29); r.cur++; return rn; } }
		seen[r.seen_len++] = rn;
		r.cur++;
		if (r.seen_len >= r.thresh && r.cur < r.n) {
			// convert to a swap list of everything not yet seen
			for (uint32_t i = 1; i < r.seen_len; i++) {       // sort seen ascending
				const uint32_t v = seen[i];
				uint32_t j = i;
				while (j > 0 && seen[j - 1] > v) { seen[j] = seen[j - 1]; j--; }
				seen[j] = v;
			}
			const uint32_t nl = r.n - r.cur;
			r.list_off = lists_alloc(nl);
			r.list_len = nl;
			BT2_G uint32_t* l = WK.lists + r.list_off;
			uint32_t prev = 0, cur = 0;
			for (uint32_t i = 0; i <= seen_sz; i++) {
				for (uint32_t j = prev; j < seen[i]; j++) l[cur++] = j;
				prev = seen[i] + 1;
			}
			for (uint32_t j = prev; j < r.n; j++) l[cur++] = j;
			r.seen_len = 0;
			r.cur = 0;
			r.n = nl;
			r.converted = 1;
			r.swaplist = 1;
		}
		return rn;
	}

	// ---------------------------------------------------------------------------------------------------------------------
	// RowSampler + Random1toN for up to 64 ranges WITHOUT MEMORY (prioritize's second phase; aligner_sw_driver.h:179-256,
	// random_util.h:32-219).  The draws have to be replayed one by one -- each consumes the read's RNG, and which range a draw
	// lands in depends on the draws before it -- but nothing about them needs a list in memory:
	//  * per range (lane j = range j, one register each): weight, running sum of the weights still in play, rows, cursor,
	//    seen count, conversion threshold, flags, first row;
	//  * a swap list (Random1toN of a range of < 128 rows, or after conversion) starts as 0, 1, 2 ... and a draw changes ONE
	//    entry that is ever read again (position rr receives the value of position cur; position cur is never looked at
	//    after the cursor has passed it).  So the list is the identity plus a sparse set of overrides;
	//  * a seen list (range of >= 128 rows) is a set of values, at most one more per draw;
	//  * the list a seen list is converted to -- every unseen value, ascending -- is  i -> i + #{seen s : s - rank(s) <= i};
	//    again no list, just the seen values with their ranks subtracted.
	// All three are entries (kind, range, position/value; payload) of ONE table kept in the lanes of kSampTabRegs register
	// pairs: a lookup is one compare + ballot per 64 entries, an insertion one v_writelane.  One entry per draw at most, so
	// the table holds every run with maxelt <= 64 * kSampTabRegs - 2 (-k <= 6); others take the arena path below.
	// (Round 3 kept running sums and Random1toN records in LDS and the lists in the arena: a draw cost ~1.4 us -- two
	// dependent LDS round trips for the pick, one for the record, a global round trip for the list, three stores for the
	// sampled row whose completion the next wait on the vector-memory counter also waits for.)
	static constexpr int kSampTabRegs = 8;
	static constexpr uint32_t kTabSwap = 1u << 30, kTabSeen = 2u << 30, kTabConv = 3u << 30, kTabNone = 0xffffffffu;
	// (entry e of the table: the platform decides where it lives -- on the device the newest entries sit in register 0 and a full register
	// is handed down the array, so that no register is ever selected by a run-time index: Plat::tab_lookup / tab_set / tab_append)
	struct SampTab { typename Plat::template LaneRegs<kSampTabRegs> k, v; uint32_t n; };
	// Draws rows until maxelt (or every element) has been drawn; appends them to the extension list (Work::srows).  Returns the
	// number of rows drawn in all (nelt_added).
	BT2_HDN uint64_t sample_rows_fast(uint32_t sai_, uint32_t n_masses_, uint64_t maxelt_, uint64_t nelt_, uint64_t nelt_added_) {
		const uint32_t sai = Plat::uni(sai_), n_masses = Plat::uni(n_masses_);
		const uint64_t maxelt = Plat::uni(maxelt_), nelt = Plat::uni(nelt_);
		uint64_t nelt_added = Plat::uni(nelt_added_);
#ifdef BT2G_SAMP_PROF
		const uint64_t tcall_ = Plat::clock();
#endif
		Rng g; g.last = Plat::uni(ST.rnd.last); g.lastOff = Plat::uni(ST.rnd.lastOff);      // (LDS loads arrive in vector registers: lane-varying to the compiler unless told otherwise)
		uint32_t n_satpos = Plat::uni(HOT.n_satpos);
		const uint32_t n_full = Plat::uni(HOT.n_satpos_full);
		BT2_G SampRow* const srows = WK.srows;
		const bool all_hits = PRM.all_hits != 0;
		typename Plat::LaneReg mlo, mhi, plo, phi, rn, rcur, rseen, rthr, rfl, tlo, thi;      // lane j: range sai + j
		typename Plat::LaneReg olo, ohi, osrc;                                                // rows drawn, 64 at a time on their way to Work::srows
		SampTab tab;
		tab.n = 0;
		Plat::tab_zero(tab.k); Plat::tab_zero(tab.v);
		Plat::lanes_zero(rcur); Plat::lanes_zero(rseen); Plat::lanes_zero(olo); Plat::lanes_zero(ohi); Plat::lanes_zero(osrc);
		Plat::lanes_zero(plo); Plat::lanes_zero(phi);
		Plat::samp_setup(&WK.satpos2[sai], n_masses, all_hits, mlo, mhi, rn, rthr, rfl, tlo, thi);
		uint64_t live = n_masses >= 64u ? ~0ull : ((1ull << n_masses) - 1ull);
		double mass = Plat::prefix_live(mlo, mhi, live, plo, phi);
		uint32_t obase = n_satpos - n_full, ocnt = 0;
		uint64_t draws = 0;
		uint32_t draws_batched = 0, nbatch_prof = 0;
		bool full = false;
		// ---- batched replay (round 6) ----
		// A draw consumes the read's RNG and decides which range the next one lands in, so the draws cannot be reordered -- but up to 64 of them can be
		// REPLAYED side by side, lane d = the d-th draw from here, as long as no draw of the batch changes what a later one sees:
		//  * the RNG is an LCG (random_source.h:34-159): the state after k steps is A_k * s + C_k (mod 2^32).  A draw takes four steps (nextFloat's
		//    nextU32, Random1toN's nextU32) unless a seen-list draw is rejected, so lane d starts from the state 4 d steps on (jA, jC below);
		//  * the range a draw picks depends on the running sums of the ranges in play: unchanged until a range is used up -- the batch is committed
		//    up to and including the draw that exhausts a range;
		//  * a draw's cursor in its range = the range's cursor + the earlier draws of the batch in the same range (occ);
		//  * what a draw reads from the table (positions cur and rr of a swap list, "seen before?" of a seen list) must not have been written by an
		//    earlier draw of the batch: the batch is committed up to the first draw with such a conflict, up to the first rejected seen-list draw
		//    and up to the first draw that converts its range (thresh reached) -- that one draw then goes through the serial code below.
		// Everything is lane code over the same registers the serial path uses; SAM parity of every differential test pins it on the CPU twin.
		constexpr uint32_t kLcgA = 1664525u, kLcgC = 1013904223u;
		constexpr uint32_t kA2 = kLcgA * kLcgA, kA4 = kA2 * kA2, kC4 = kLcgC * (kLcgA * kA2 + kA2 + kLcgA + 1u);
		constexpr uint32_t kBatchMin = 6u;
		typename Plat::LaneReg jA, jC;
		BT2_FOR_LANES(l) {
			uint32_t A = 1u, C = 0u, pa = kA4, pc = kC4;
			for (uint32_t b = 0; b < 6u; b++) {
				if ((l >> b) & 1u) { C = C * pa + pc; A = A * pa; }
				pc = pc * pa + pc; pa = pa * pa;
			}
			LV(jA) = A; LV(jC) = C;
		}
#ifdef BT2G_SAMP_PROF
		// experiment builds: device clocks of the batch's parts, booked under phase slots that are otherwise (nearly) empty on unpaired end-to-end batches
#define SAMP_T(slot) do { const uint64_t t__ = Plat::clock(); HOT.t_phase[slot] += t__ - tsp_; tsp_ = t__; } while (0)
		uint64_t tsp_ = Plat::clock();
		HOT.t_phase[20] += tsp_ - tcall_;
#else
#define SAMP_T(slot) do {} while (0)
#endif
		bool serial_next = false;
		// The batch looks table entries up by key in hash tables that mirror the register table -- S: the seen-list keys (a set), W: swap-list key ->
		// entry index -- and finds the draws of a batch that meet in one position through a third, per-batch table Bt (key -> the lanes writing it).
		// They live in the launch's dynamic LDS, which holds DP-window state otherwise and is free while rows are sampled (Plat::sh_begin: slots of S
		// and W; 0 when the launch has no room: no batching then).  A table is kept while it is at most 3/4 full; past that the call goes on serially.
#ifdef BT2G_SAMP_SERIAL
		const uint32_t shN = 0u;      // A/B builds: every draw through the serial path below
#else
		const uint32_t shN = (maxelt < nelt ? maxelt : nelt) - nelt_added >= (uint64_t)kBatchMin ? Plat::uni(Plat::sh_begin()) : 0u;
#endif
		const uint32_t capS = (shN & 0xffffu) - ((shN & 0xffffu) >> 2), capW = (shN >> 16) - ((shN >> 16) >> 2);
		uint32_t nS_ent = 0, nW_ent = 0;      // entries in S / W
		bool sh_live = shN != 0u;             // S and W still take entries
		uint32_t n_mir = 0xffffffffu;         // once one of them is full: entries of the register table from this index on are not in S / W (the batch scans them)
		while (nelt_added < maxelt && nelt_added < nelt) {
			if (shN != 0u && !serial_next) {
				const uint64_t want = (maxelt < nelt ? maxelt : nelt) - nelt_added;
				uint32_t B = want > 64ull ? 64u : (uint32_t)want;
				{ const uint32_t room_rows = (uint32_t)kMaxSatpos - n_satpos, room_tab = 64u * (uint32_t)kSampTabRegs - 2u > tab.n ? 64u * (uint32_t)kSampTabRegs - 2u - tab.n : 0u;
				  if (B > room_rows) B = room_rows;
				  if (B > room_tab) B = room_tab;
				  if (sh_live && (nS_ent + B > capS || nW_ent + B > capW)) { sh_live = false; n_mir = tab.n; } }
				if (B >= kBatchMin && live != 0ull) {
#ifdef BT2G_SAMP_PROF
					tsp_ = Plat::clock();
#endif
					// (1) every lane's RNG draws and its point on the running sums
					typename Plat::LaneReg u2r, x4r, rdlo, rdhi;
					const uint32_t g0 = g.last;
					BT2_FOR_LANES(l) {
						uint32_t x = LV(jA) * g0 + LV(jC);
						x = kLcgA * x + kLcgC; uint32_t r1 = x >> 16; x = kLcgA * x + kLcgC; r1 ^= x;
						x = kLcgA * x + kLcgC; uint32_t r2 = x >> 16; x = kLcgA * x + kLcgC; r2 ^= x;
						const float f = (float)r1 / (float)0xffffffffu;
						const double rdv = (double)(f * mass);
						uint64_t u; __builtin_memcpy(&u, &rdv, 8);
						LV(u2r) = r2; LV(x4r) = x; LV(rdlo) = (uint32_t)u; LV(rdhi) = (uint32_t)(u >> 32);
					}
					// (2) RowSampler::next of every lane + how many earlier lanes picked the same range; lane i of mblo/mbhi: the lanes that picked range i
					typename Plat::LaneReg pick, occ, mblo, mbhi, nw;
					Plat::lanes_zero(pick); Plat::lanes_zero(occ); Plat::lanes_zero(mblo); Plat::lanes_zero(mbhi);
					uint64_t rem = B >= 64u ? ~0ull : ((1ull << B) - 1ull);
					const uint32_t last_live = 63u - (uint32_t)__builtin_clzll(live);
					for (uint64_t lvm = live; lvm != 0ull && rem != 0ull; lvm &= lvm - 1ull) {
						const uint32_t i = (uint32_t)__builtin_ctzll(lvm);
						const double pfx = f64_of(Plat::lane(plo, i), Plat::lane(phi, i));
						BT2_FOR_LANES(l) { LV(nw) = (((rem >> l) & 1ull) && (i == last_live || f64_of(LV(rdlo), LV(rdhi)) < pfx)) ? 1u : 0u; }
						const uint64_t m = Plat::ballot(nw);
						if (m) {
							BT2_FOR_LANES(l) { if (LV(nw)) { LV(pick) = i; LV(occ) = (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull)); } }
							Plat::set_lane(mblo, i, (uint32_t)m); Plat::set_lane(mbhi, i, (uint32_t)(m >> 32));
							rem &= ~m;
						}
					}
					SAMP_T(0);
					// (3) the state of each lane's range, (4) its Random1toN draw and the table keys it has to look at
					const typename Plat::LaneReg gn = Plat::gather(rn, pick), gc = Plat::gather(rcur, pick), gf = Plat::gather(rfl, pick), gs = Plat::gather(rseen, pick),
						gt = Plat::gather(rthr, pick), gml = Plat::gather(mblo, pick), gmh = Plat::gather(mbhi, pick);
					typename Plat::LaneReg curr, posr, keyA, keyB, convk, bad, exh, fa, va, fb, vb, eb, ca, cb;
					BT2_FOR_LANES(l) {
						const uint32_t n = LV(gn), c = LV(gc) + LV(occ), fl = LV(gf);
						uint32_t bd = 0, ex = 0, pos = 0, kA = 0, kB = 0, kC = 0;
						if (l < B) {
							const uint32_t rkey = LV(pick) << 24;
							if (c >= n) bd = 1u;
							else if (fl & 1u) {
								if (n == 1u && c == 0u && !(fl & 2u)) bd = 1u;
								else { pos = c + LV(u2r) % (n - c); kA = kTabSwap | rkey | c; kB = kTabSwap | rkey | pos; if (fl & 2u) kC = kTabConv | rkey; }
							} else {
								pos = LV(u2r) % n; kB = kTabSeen | rkey | pos;
								if (LV(gs) + LV(occ) + 1u >= LV(gt) && c + 1u < n) bd = 1u;      // this draw converts the range: serial
							}
							if (!bd && c + 1u >= n) ex = 1u;
						}
						LV(curr) = c; LV(posr) = pos; LV(keyA) = kA; LV(keyB) = kB; LV(convk) = kC; LV(bad) = bd; LV(exh) = ex;
						LV(fa) = LV(va) = LV(fb) = LV(vb) = LV(eb) = LV(ca) = LV(cb) = 0u;
					}
					// (5) what the table says about positions cur and rr (swap list) / "drawn before?" (seen list): T gives the entry, the register table its word
					{
						typename Plat::LaneReg fsw, fany, ia, ib;
						BT2_FOR_LANES(l) { LV(fsw) = (LV(keyA) != 0u) ? 1u : 0u; LV(fany) = (LV(keyB) != 0u && !LV(bad)) ? 1u : 0u; }
						Plat::sh_get(shN, keyA, fsw, fa, ia);
						Plat::sh_get(shN, keyB, fany, fb, ib);
						const typename Plat::LaneReg ga_ = Plat::tab_gather(tab.v, ia), gb_ = Plat::tab_gather(tab.v, ib);
						BT2_FOR_LANES(l) { LV(va) = LV(ga_); LV(vb) = LV(gb_); LV(eb) = LV(ib); }
						if (tab.n > n_mir) {
							// entries S / W had no room for: one pass over them
							Plat::tab_for_each(tab.k, tab.v, tab.n, [&](uint32_t e, uint32_t ke, uint32_t ve) {
								BT2_FOR_LANES(l) {
									if (ke == LV(keyA)) { LV(fa) = 1u; LV(va) = ve; }
									if (ke == LV(keyB)) { LV(fb) = 1u; LV(vb) = ve; LV(eb) = e; }
								}
							}, n_mir);
						}
						// converted ranges (rare): position i of the list is i + the number of seen values with value - rank <= i -- a pass over the table
						typename Plat::LaneReg cvf;
						BT2_FOR_LANES(l) { LV(cvf) = LV(convk) != 0u ? 1u : 0u; }
						if (Plat::ballot(cvf)) {
							Plat::tab_for_each(tab.k, tab.v, tab.n, [&](uint32_t e, uint32_t ke, uint32_t ve) {
								(void)e;
								BT2_FOR_LANES(l) { if ((ke & 0xff000000u) == LV(convk)) { if (ve <= LV(curr)) LV(ca) = LV(ca) + 1u; if (ve <= LV(posr)) LV(cb) = LV(cb) + 1u; } }
							});
						}
					}
					// a seen-list draw whose value is in the table is rejected (drawn again): serial
					BT2_FOR_LANES(l) { if (l < B && !(LV(gf) & 1u) && LV(fb)) LV(bad) = 1u; }
					SAMP_T(1);
					// (6) draws of the batch that meet in one position.  Every draw that writes -- a seen-list draw its value, a swap-list draw position rr
					//     (unless rr = cur) -- joins, in Bt, the set of lanes writing that key.  A seen-list draw that finds an earlier lane under its value is
					//     a rejection (serial).  A swap-list draw reads positions cur and rr: what the LATEST earlier draw s of the batch put there is what s
					//     found at ITS cur (depA / depB); of several writes to one position the last committed one is the table's entry (wmlo/wmhi: the writers).
					typename Plat::LaneReg depA, depB, wrote, wmlo, wmhi;
					{
						typename Plat::LaneReg wr, fsw, fany, alo, ahi;
						BT2_FOR_LANES(l) {
							const uint32_t ok = (l < B && !LV(bad)) ? 1u : 0u, sw = LV(gf) & 1u;
							LV(wrote) = (ok && sw && LV(posr) != LV(curr)) ? 1u : 0u;
							LV(wr) = (ok && (!sw || LV(posr) != LV(curr))) ? 1u : 0u;
							LV(fsw) = (ok && sw) ? 1u : 0u; LV(fany) = ok;
						}
						Plat::bh_clear(shN);
						Plat::bh_mark(shN, keyB, wr);
						Plat::bh_get(shN, keyA, fsw, alo, ahi);
						Plat::bh_get(shN, keyB, fany, wmlo, wmhi);
						BT2_FOR_LANES(l) {
							uint32_t da = 0xffffffffu, db = 0xffffffffu;
							if (LV(fany)) {
								const uint64_t below = (1ull << l) - 1ull;
								const uint64_t ma = ((((uint64_t)LV(ahi)) << 32) | (uint64_t)LV(alo)) & below, mb = ((((uint64_t)LV(wmhi)) << 32) | (uint64_t)LV(wmlo)) & below;
								if (LV(gf) & 1u) {
									if (ma) da = 63u - (uint32_t)__builtin_clzll(ma);
									if (mb) db = 63u - (uint32_t)__builtin_clzll(mb);
								} else if (mb) LV(bad) = 1u;
							}
							LV(depA) = da; LV(depB) = db;
						}
					}
					SAMP_T(15);
					// (7) the batch is good up to the first rejection / conversion, and up to and including the first draw that uses its range up
					const uint64_t badm = Plat::ballot(bad), exm = Plat::ballot(exh);
					uint32_t L = B;
					if (badm) { const uint32_t fbad = (uint32_t)__builtin_ctzll(badm); if (fbad < L) L = fbad; }
					bool used_up = false;
					if (exm) { const uint32_t fex = (uint32_t)__builtin_ctzll(exm); if (fex + 1u <= L) { L = fex + 1u; used_up = true; } }
					if (L < B && !used_up) serial_next = true;      // the draw after the batch needs the serial path
#ifdef BT2G_SAMP_STATS
					{ static unsigned long st_[6] = {0,0,0,0,0,0}, nb_ = 0; nb_++;
					  if (L == B) st_[0]++; else if (used_up) st_[1]++; else {
					    const uint32_t fb_ = (uint32_t)__builtin_ctzll(badm);
					    const uint32_t fl_ = Plat::lane(gf, fb_), c_ = Plat::lane(curr, fb_), n_ = Plat::lane(gn, fb_);
					    if (c_ >= n_) st_[5]++; else if (!(fl_ & 1u) && Plat::lane(fb, fb_)) st_[2]++; else if (!(fl_ & 1u) && Plat::lane(gs, fb_) + Plat::lane(occ, fb_) + 1u >= Plat::lane(gt, fb_) && c_ + 1u < n_) st_[3]++; else st_[4]++; }
					  if ((nb_ & (nb_ - 1)) == 0 || getenv("BT2G_SAMP_STATS_ALL")) fprintf(stderr, "SAMPSTAT batches %lu full %lu used_up %lu rejected %lu convert %lu seen_twice %lu past_end %lu  (B %u L %u live %d)\n", nb_, st_[0], st_[1], st_[2], st_[3], st_[4], st_[5], B, L, __builtin_popcountll(live)); }
#endif
					if (L > 0u) {
						// (8) commit draws 0 .. L-1
						if (ocnt > 0u) { Plat::flush_samp_rows(srows + obase, olo, ohi, osrc, ocnt); obase += ocnt; ocnt = 0u; }
						const uint64_t cm = L >= 64u ? ~0ull : ((1ull << L) - 1ull);
						// what each swap-list draw finds at its position cur: the table's (or the identity's) word, or what an earlier draw of the batch put there --
						// which is what THAT draw found at its own cur: chains resolve front to back, one link per round
						typename Plat::LaneReg av, rsv;
						BT2_FOR_LANES(l) {
							const uint32_t fl = LV(gf), c = LV(curr);
							LV(av) = LV(fa) ? LV(va) : ((fl & 2u) ? c + LV(ca) : c);
							LV(rsv) = (l >= L || LV(depA) == 0xffffffffu) ? 1u : 0u;
						}
						for (;;) {
							typename Plat::LaneReg unres;
							BT2_FOR_LANES(l) { LV(unres) = LV(rsv) ? 0u : 1u; }
							if (!Plat::ballot(unres)) break;
							typename Plat::LaneReg dix;
							BT2_FOR_LANES(l) { LV(dix) = LV(depA) & 63u; }
							const typename Plat::LaneReg g_r = Plat::gather(rsv, dix), g_a = Plat::gather(av, dix);
							BT2_FOR_LANES(l) { if (!LV(rsv) && LV(g_r)) { LV(av) = LV(g_a); LV(rsv) = 1u; } }
						}
						typename Plat::LaneReg app, akey, aval, setf, wlo, whi, wsrc, dixb;
						BT2_FOR_LANES(l) { LV(dixb) = LV(depB) & 63u; }
						const typename Plat::LaneReg g_b = Plat::gather(av, dixb);
						const typename Plat::LaneReg gtl = Plat::gather(tlo, pick), gth = Plat::gather(thi, pick);
						BT2_FOR_LANES(l) {
							const uint32_t fl = LV(gf), c = LV(curr), pos = LV(posr);
							uint32_t ret = 0, ap = 0, sf = 0, avv = 0;
							if (l < L) {
								if (fl & 1u) {
									const uint32_t a = LV(av);
									if (pos == c) ret = a;
									else {
										ret = LV(depB) != 0xffffffffu ? LV(g_b) : (LV(fb) ? LV(vb) : ((fl & 2u) ? pos + LV(cb) : pos));
										avv = a;
										// of several writes of the batch to this position the last committed one is the entry
										const uint64_t later = (((((uint64_t)LV(wmhi)) << 32) | (uint64_t)LV(wmlo)) & cm) >> l >> 1;
										if (!later) { if (LV(fb)) sf = 1u; else ap = 1u; }
									}
								} else { ret = pos; ap = 1u; avv = 0u; }
							}
							const uint64_t topf = (((uint64_t)LV(gth) << 32) | (uint64_t)LV(gtl)) + (uint64_t)ret;
							LV(wlo) = (uint32_t)topf; LV(whi) = (uint32_t)(topf >> 32); LV(wsrc) = LV(pick) + sai;
							LV(app) = ap; LV(akey) = LV(keyB); LV(aval) = avv; LV(setf) = sf;
						}
						for (uint64_t sm = Plat::ballot(setf); sm != 0ull; sm &= sm - 1ull) {      // position rr already has an entry: it takes the value of position cur
							const uint32_t ls = (uint32_t)__builtin_ctzll(sm);
							Plat::tab_set_at(tab.v, Plat::lane(eb, ls), Plat::lane(aval, ls));
						}
						{
							// new entries: register table, and T (key -> entry index)
							const uint64_t am = Plat::ballot(app);
							typename Plat::LaneReg aidx;
							BT2_FOR_LANES(l) { LV(aidx) = tab.n + (uint32_t)__builtin_popcountll(am & ((1ull << l) - 1ull)); }
							if (sh_live) Plat::sh_put(shN, akey, aidx, app);
							Plat::tab_append_lanes(tab.k, tab.v, tab.n, app, akey, aval);
							{ typename Plat::LaneReg asn; BT2_FOR_LANES(l) { LV(asn) = (LV(app) && (LV(akey) >> 30) == 2u) ? 1u : 0u; }
							  const uint32_t ns_ = (uint32_t)__builtin_popcountll(Plat::ballot(asn)); nS_ent += ns_; nW_ent += (uint32_t)__builtin_popcountll(am) - ns_; }
						}
						Plat::flush_samp_rows(srows + obase, wlo, whi, wsrc, L);
						obase += L; n_satpos += L; nelt_added += (uint64_t)L;
						// cursors (and seen counts) of the ranges
						BT2_FOR_LANES(l) {
							const uint32_t cnt = (uint32_t)__builtin_popcountll((((uint64_t)LV(mbhi) << 32) | (uint64_t)LV(mblo)) & cm);
							LV(rcur) = LV(rcur) + cnt;
							if (!(LV(rfl) & 1u)) LV(rseen) = LV(rseen) + cnt;
						}
						draws += (uint64_t)L;
						g.last = Plat::lane(x4r, L - 1u); g.lastOff = 0u;
						if (used_up) {
							const uint32_t pe = Plat::lane(pick, L - 1u);
							live &= ~(1ull << pe);
							mass -= f64_of(Plat::lane(mlo, pe), Plat::lane(mhi, pe));
							Plat::prefix_live(mlo, mhi, live, plo, phi);
						}
						nbatch_prof += 0x10000u; draws_batched += L;      // profile: batches, draws committed by batches
						SAMP_T(16);
						continue;
					}
				}
			}
			serial_next = false;
#ifdef BT2G_SAMP_PROF
			struct SerT { uint64_t t0; BT2_HD ~SerT() { Plat::hot().t_phase[10] += Plat::clock() - t0; } } sert_{Plat::clock()};
#endif
			// RowSampler::next: first range still in play whose running sum exceeds rd, else the last one in play
			const double rd = (double)(g.nextFloat() * mass);
			const uint32_t pick = Plat::pick_prefix(plo, phi, live, rd);
			uint32_t n = Plat::lane(rn, pick), cur = Plat::lane(rcur, pick), fl = Plat::lane(rfl, pick);      // fl: bit 0 swap list, bit 1 converted
			const uint32_t rkey = pick << 24;
			draws += 1ull;
			uint32_t ret;
			// Random1toN::next
			if (fl & 1u) {
				if (cur == 0 && !(fl & 2u) && n == 1) { cur = 1; ret = 0; }
				else {
					const uint32_t rr = cur + (g.nextU32() % (n - cur));
					uint32_t a, b;
					if (!Plat::tab_lookup(tab.k, tab.v, tab.n, kTabSwap | rkey | cur, a))
						a = (fl & 2u) ? cur + Plat::tab_count_le(tab.k, tab.v, tab.n, kTabConv | rkey, cur) : cur;
					if (rr == cur) b = a;
					else {
						const bool have = Plat::tab_lookup(tab.k, tab.v, tab.n, kTabSwap | rkey | rr, b);
						if (have) Plat::tab_set(tab.k, tab.v, tab.n, kTabSwap | rkey | rr, a);      // (position cur gets b, and is never read again)
						else {
							b = (fl & 2u) ? rr + Plat::tab_count_le(tab.k, tab.v, tab.n, kTabConv | rkey, rr) : rr;
							Plat::tab_append(tab.k, tab.v, tab.n, kTabSwap | rkey | rr, a);
							if (sh_live) { if (nW_ent < capW) { Plat::sh_put1(shN, kTabSwap | rkey | rr, tab.n - 1u); nW_ent++; } else { sh_live = false; n_mir = tab.n - 1u; } }
						}
					}
					cur++;
					ret = b;
				}
			} else {
				uint32_t rnv;
				uint32_t dummy_;
				do { rnv = g.nextU32() % n; } while (Plat::tab_lookup(tab.k, tab.v, tab.n, kTabSeen | rkey | rnv, dummy_));
				ret = rnv;
				uint32_t seen = Plat::lane(rseen, pick);
				Plat::tab_append(tab.k, tab.v, tab.n, kTabSeen | rkey | rnv, 0u);
				if (sh_live) { if (nS_ent < capS) { Plat::sh_put1(shN, kTabSeen | rkey | rnv, tab.n - 1u); nS_ent++; } else { sh_live = false; n_mir = tab.n - 1u; } }
				seen++; cur++;
				if (seen >= Plat::lane(rthr, pick) && cur < n) {
					// convert to a swap list of everything not yet seen, ascending (random_util.h:133-158): the seen values stay in the
					// table, each with its rank subtracted.  Not rare on repeats: a 200-copy family hit by every seed of a read crosses
					// thresh = 20 in each of its ranges.
					Plat::tab_convert(tab.k, tab.v, tab.n, pick);
					n -= cur; cur = 0; seen = 0; fl = 3u;
					Plat::set_lane(rn, pick, n); Plat::set_lane(rfl, pick, fl);
				}
				Plat::set_lane(rseen, pick, seen);
			}
			Plat::set_lane(rcur, pick, cur);
			if (n > 0 && cur >= n) {      // the range is used up: out of the sampler
				live &= ~(1ull << pick);
				mass -= f64_of(Plat::lane(mlo, pick), Plat::lane(mhi, pick));
				Plat::prefix_live(mlo, mhi, live, plo, phi);
			}
			if (n_satpos >= (uint32_t)kMaxSatpos) { full = true; break; }
			{
				const uint64_t topf = (((uint64_t)Plat::lane(thi, pick) << 32) | (uint64_t)Plat::lane(tlo, pick)) + (uint64_t)ret;
				Plat::set_lane(olo, ocnt, (uint32_t)topf); Plat::set_lane(ohi, ocnt, (uint32_t)(topf >> 32)); Plat::set_lane(osrc, ocnt, pick + sai);
				ocnt++; n_satpos++;
				if (ocnt == 64u) { Plat::flush_samp_rows(srows + obase, olo, ohi, osrc, 64u); obase += 64u; ocnt = 0; }
			}
			nelt_added++;
			if (tab.n + 2u > 64u * (uint32_t)kSampTabRegs) { full = true; break; }      // (cannot happen: the caller checked maxelt against the table)
		}
		if (ocnt > 0) Plat::flush_samp_rows(srows + obase, olo, ohi, osrc, ocnt);
#ifdef BT2G_SAMP_STATS
		{ // per call: entries by kind, bytes a dense form of the small ranges' lists would take
		  uint32_t nseen = 0, nsw_small = 0, nsw_conv = 0, nconv = 0, dense = 0;
		  Plat::tab_for_each(tab.k, tab.v, tab.n, [&](uint32_t, uint32_t ke, uint32_t) {
		    const uint32_t kind = ke >> 30, rg = (ke >> 24) & 63u;
		    if (kind == 2) nseen++; else if (kind == 3) nconv++; else if (kind == 1) { if (Plat::lane(rfl, rg) & 2u) nsw_conv++; else nsw_small++; } });
		  for (uint32_t i = 0; i < n_masses; i++) { const uint32_t sz = WK.satpos2[sai + i].size; if (sz < 128u) dense += sz; }
		  static unsigned long calls_ = 0, h_[5][8]; calls_++;
		  auto bk = [](uint32_t v) { return v == 0 ? 0 : v <= 16 ? 1 : v <= 32 ? 2 : v <= 64 ? 3 : v <= 128 ? 4 : v <= 256 ? 5 : v <= 512 ? 6 : 7; };
		  h_[0][bk(nseen)]++; h_[1][bk(nsw_small)]++; h_[2][bk(nsw_conv)]++; h_[3][bk(dense)]++; h_[4][bk((uint32_t)draws)]++;
		  if ((calls_ & (calls_ - 1)) == 0) { const char* nm[5] = {"seen", "swap_small", "swap_conv", "dense_bytes", "draws"};
		    for (int a = 0; a < 5; a++) { fprintf(stderr, "SAMPHIST calls %lu %-12s 0:%lu <=16:%lu <=32:%lu <=64:%lu <=128:%lu <=256:%lu <=512:%lu more:%lu\n", calls_, nm[a], h_[a][0], h_[a][1], h_[a][2], h_[a][3], h_[a][4], h_[a][5], h_[a][6], h_[a][7]); } } }
#endif
		ST.rnd = g; HOT.mass = mass; HOT.n_satpos = n_satpos;
		HOT.t_phase[21] += draws | ((uint64_t)draws_batched << 32);      // profile: draws | draws committed by batches << 32
		HOT.n_dp_pass += nbatch_prof;                                      // (high half: batches)
		if (full) ovf(13);
		return nelt_added;
	}

	// Entry i of the extension list for the consumer loops: whole records as they are, sampled rows expanded into WK.sp_view
	// (satpos_commit() remembers that the row has been taken)
	BT2_HD bool satpos_taken(uint32_t i) const { return i >= HOT.n_satpos_full && WK.srows[i - HOT.n_satpos_full].done != 0; }
	BT2_HD BT2_G SatPos& satpos_view(uint32_t i) {
		if (i < HOT.n_satpos_full) return WK.satpos[i];
		const SampRow sr = WK.srows[i - HOT.n_satpos_full];
		BT2_G SatPos& s = WK.sp_view;
		Plat::copy_words(&s, &WK.satpos2[sr.src], (uint32_t)(sizeof(SatPos) / 4));
		s.topf = sr.topf; s.topb = (uint64_t)kOffMask; s.size = 1;
		r1n_init(s.rnd, 1, PRM.all_hits != 0);
		return s;
	}
	BT2_HD void satpos_commit(uint32_t i, const BT2_G SatPos& sp) { if (i >= HOT.n_satpos_full) WK.srows[i - HOT.n_satpos_full].done = r1n_done(sp.rnd) ? 1u : 0u; }

	// =================================================================================
	// C. seed-hit extension bookkeeping
	// =================================================================================
	// SwDriver::extend (aligner_sw_driver.cpp:299-484)
	BT2_HDN void extend_hit(TOff topf, TOff botf, TOff topb, TOff botb, bool fw, uint32_t off, uint32_t len,
	                       uint32_t& nlex, uint32_t& nrex) {
		FmCount cnt; cnt.bwops = 0; cnt.sides = 0;
		HotRd rd;
		fm_extend_hit(IX, rd, HOT.len, topf, botf, topb, botb, fw, off, len, nlex, nrex, cnt, (PRM.do_extend & 2) == 0);
		HOT.n_bwops_ext += cnt.bwops; HOT.n_sides += cnt.sides;
#ifdef BT2G_CHECK_EXTEND_TEXT
		// test builds: the text-comparison form the batch kernel uses for hits of up to kExtRows rows must agree with the LF walk
		if (botf - topf >= 1 && (uint64_t)(botf - topf) <= kExtRows) {
			uint64_t p_[kExtRows]; bool in_text = true;
			for (uint32_t k = 0; k < kExtRows; k++) { p_[k] = 0; if (k < (uint32_t)(botf - topf)) { uint32_t st_ = 0; p_[k] = (uint64_t)get_offset(IX.fw, (TOff)(topf + k), st_); if (p_[k] >= (uint64_t)IX.fw.len) in_text = false; } }
			if (in_text) {
				uint32_t l2 = 0, r2 = 0, wk = 0;
				fm_extend_rows_text(IX, rd, HOT.len, p_, (uint32_t)(botf - topf), fw, off, len, l2, r2, wk, (PRM.do_extend & 2) == 0);
				static unsigned long n_ = 0, nm_ = 0; n_++; if (botf - topf > 1) nm_++;
				if (l2 != nlex || r2 != nrex) { fprintf(stderr, "extend mismatch: rows %u LF %u/%u text %u/%u (fw %d off %u len %u)\n", (unsigned)(botf - topf), nlex, nrex, l2, r2, (int)fw, off, len); abort(); }
				if ((n_ & (n_ - 1)) == 0) fprintf(stderr, "extend text form checked on %lu hits (%lu with several rows)\n", n_, nm_);
			}
		}
#endif
	}

	// SATupleAndPos::operator< (aligner_sw_driver.h:150-160)
	template <typename SA, typename SB> BT2_HD static bool satpos_less(const SA& a, const SB& o) {
		if (a.size < o.size) return true;
		if (a.size > o.size) return false;
		if (a.topf < o.topf) return true;
		if (a.topf > o.topf) return false;
		if (a.offidx < o.offidx) return true;
		if (a.offidx > o.offidx) return false;
		if (a.rdoff < o.rdoff) return true;
		if (a.rdoff > o.rdoff) return false;
		if (a.seedlen < o.seedlen) return true;
		if (a.seedlen > o.seedlen) return false;
		if (a.fw && !o.fw) return true;
		return false;
	}

	// SwDriver::eeSaTups (aligner_sw_driver.cpp:66-291)
	BT2_HDN void ee_sa_tups(uint64_t& nelt_out, uint64_t maxelt_) {
		const uint64_t maxelt = Plat::uni(maxelt_);
		HOT.n_satpos = 0;
		HOT.lists_used = 0;
		nelt_out = 0;
		const uint64_t szfw = HOT.exact[0].bot - HOT.exact[0].top, szrc = HOT.exact[1].bot - HOT.exact[1].top;
		const uint64_t tot = szfw + szrc;
		bool done = false;
		auto add = [&](const EEHit& hit, int ee_idx, uint64_t top, uint64_t width) {
			if (HOT.n_satpos >= (uint32_t)kMaxSatpos) { ovf(8); done = true; return; }
			BT2_G SatPos& s = WK.satpos[HOT.n_satpos++];
			s.topf = top; s.topb = (uint64_t)kOffMask; s.size = (uint32_t)width; s.orig_sz = (uint32_t)width;
			s.fw = hit.fw; s.offidx = 0; s.rdoff = 0; s.seedlen = HOT.len; s.nlex = s.nrex = 0;
			s.ee = ee_idx;
			r1n_init(s.rnd, (uint32_t)width, PRM.all_hits != 0);
			nelt_out += width;
			if (nelt_out >= maxelt) done = true;
		};
		auto add_trimmed = [&](const EEHit& hit, int ee_idx) {
			uint64_t tops[2] = {hit.top, 0}, bots[2] = {hit.bot, 0};
			const uint64_t width = hit.bot - hit.top;
			if (nelt_out + width > maxelt) {
				const uint64_t trim = (nelt_out + width) - maxelt;
				const uint64_t rn = (PRM.large_index ? ST.rnd.nextU64() : (uint64_t)ST.rnd.nextU32()) % width;
				const uint64_t newwidth = width - trim;
				if (hit.top + rn + newwidth > hit.bot) {
					tops[0] = hit.top + rn; bots[0] = hit.bot;
					tops[1] = hit.top; bots[1] = hit.top + newwidth - (bots[0] - tops[0]);
				} else {
					tops[0] = hit.top + rn; bots[0] = tops[0] + newwidth;
				}
			}
			for (int i = 0; i < 2 && !done; i++) {
				if (bots[i] <= tops[i]) break;
				add(hit, ee_idx, tops[i], bots[i] - tops[i]);
			}
		};
		if (tot > 0) {
			bool fw_first = true;
			const uint64_t rn = (PRM.large_index ? ST.rnd.nextU64() : (uint64_t)ST.rnd.nextU32()) % tot;
			if (rn >= szfw) fw_first = false;
			for (int fwi = 0; fwi < 2 && !done; fwi++) {
				const bool fw = ((fwi == 0) == fw_first);
				const EEHit& hit = HOT.exact[fw ? 0 : 1];
				if (hit.bot <= hit.top) continue;
				add_trimmed(hit, fw ? -2 : -3);     // -2/-3: exact fw / rc hit
			}
		}
		if (!done && HOT.n_mm1 > 0) {
			// sort1mmEe: stable sort by score descending, then shuffle equal-score streaks (aligner_seed.h:1223)
			for (uint32_t i = 1; i < HOT.n_mm1; i++) {
				const EEHit v = WK.mm1[i];
				uint32_t j = i;
				while (j > 0 && WK.mm1[j - 1].score < v.score) { WK.mm1[j] = WK.mm1[j - 1]; j--; }
				WK.mm1[j] = v;
			}
			auto shuffle = [&](uint32_t begin, uint32_t num) {
				if (num < 2) return;
				uint32_t left = num;
				for (uint32_t i = begin; i < begin + num - 1; i++) {
					const uint64_t rndi = ST.rnd.nextU64() % left;
					if (rndi > 0) { const EEHit tmp = WK.mm1[i]; WK.mm1[i] = WK.mm1[i + rndi]; WK.mm1[i + rndi] = tmp; }
					left--;
				}
			};
			uint32_t streak = 0;
			for (uint32_t i = 1; i < HOT.n_mm1; i++) {
				if (WK.mm1[i].score == WK.mm1[i - 1].score) {
					if (streak == 0) streak = 1;
					streak++;
				} else {
					if (streak > 1) shuffle(i - streak, streak);
					streak = 0;
				}
			}
			if (streak > 1) shuffle(HOT.n_mm1 - streak, streak);
			for (uint32_t i = 0; i < HOT.n_mm1 && !done; i++) add_trimmed(WK.mm1[i], (int)i);
		}
		HOT.n_satpos_full = HOT.n_resolved = HOT.n_satpos;
	}

	BT2_HD EEHit ee_hit(int idx) const { if (idx == -2) return HOT.exact[0]; if (idx == -3) return HOT.exact[1]; return WK.mm1[idx]; }      // (a copy: the exact hits live in LDS, the 1-mismatch hits in HBM)

	// SwDriver::prioritizeSATupsRands (aligner_sw_driver.cpp:492-738)
	BT2_HDN void prioritize(int seedmms_, uint64_t maxelt_, uint64_t& nelt_out) {
		const int seedmms = Plat::uni(seedmms_);
		const uint64_t maxelt = Plat::uni(maxelt_);
		const uint32_t nsm = 5;
		HOT.n_satpos = 0; HOT.n_satpos2 = 0; HOT.lists_used = 0;
		uint64_t nrange = 0, nelt = 0, nsmall = 0, nsmall_elts = 0;
		bool big_range = false;      // a range of 2^24 rows or more (only with an enlarged --seed-cache-sz): its positions do not fit the sampler's table keys
		const uint64_t th_ = now();
		for (uint32_t i = 0; i < HOT.n_rank; i++) {
			const bool fw = HOT.rank_fw[i] != 0;
			const uint32_t offidx = HOT.rank_offs[i];
			const uint32_t rdoff = HOT.off_idx2off[offidx];
			const uint32_t seedlen = RPR.seedlen < (int32_t)HOT.len ? (uint32_t)RPR.seedlen : HOT.len;
			const HotHit& h = HOT.hits[fw ? 0 : 1][offidx];
			const uint32_t nr_here = seedmms > 0 ? (uint32_t)h.topb : 1u;      // ca.queryQval: one SATuple per reference string
			for (uint32_t ri = 0; ri < nr_here; ri++) {
			uint64_t h_topf = h.topf, h_topb = h.topb, sz = h.esize;      // the range as the seed cache holds it
			if (seedmms > 0) { const BT2_G SeedRange& sr = WK.sranges[h.topf + ri]; h_topf = sr.topf; h_topb = sr.topb; sz = sr.esize; }      // (queryQval: what the cache holds)
			if (sz == 0) continue;
			nrange++; nelt += sz;
			if (seedmms == 0) {
				const bool m2 = PRM.paired && HOT.pe.cur == 1;     // seedExRangeFw_[matei] / seedExRangeRc_[matei]
				const Work::ExtRange* range = m2 ? (fw ? WK.ex_fw2 : WK.ex_rc2) : (fw ? WK.ex_fw : WK.ex_rc);
				const uint32_t nr = m2 ? (fw ? HOT.pe.n_ex_fw2 : HOT.pe.n_ex_rc2) : (fw ? HOT.n_ex_fw : HOT.n_ex_rc);
				bool skip = false;
				for (uint32_t k = 0; k < nr; k++) {
					if (range[k].off <= rdoff && range[k].off + range[k].len >= rdoff + seedlen) {
						if (sz <= range[k].sz) { skip = true; break; }
					}
				}
				if (skip) { nrange--; nelt -= sz; continue; }
			}
			if (HOT.n_satpos2 >= (uint32_t)kMaxSat2) { ovf(9); break; }
			BT2_G SatPos& s = WK.satpos2[HOT.n_satpos2++];
			s.topf = h_topf; s.topb = h_topb; s.size = (uint32_t)sz; s.orig_sz = (uint32_t)sz;
			s.fw = fw ? 1 : 0; s.offidx = offidx; s.rdoff = rdoff; s.seedlen = seedlen; s.ee = -1;
			if (sz <= nsm) { nsmall++; nsmall_elts += sz; }
			if (sz >= (1ull << 24)) big_range = true;
			uint32_t nlex = 0, nrex = 0;
			if (PRM.do_extend) {
				if (ST.ext_pre && seedmms == 0 && h.esize == h.size) {      // (a range the cache cut short is extended as the shorter range)
					const uint32_t e = ST.pre_ext_cur[((uint64_t)ST.ridx * 2 + (fw ? 0 : 1)) * PRE->max_seeds + offidx];
					nlex = e & 0xffffu; nrex = e >> 16;
				} else extend_hit((TOff)h_topf, (TOff)(h_topf + sz), (TOff)h_topb, (TOff)(h_topb + sz), fw, rdoff, seedlen, nlex, nrex);
			}
			s.nlex = nlex; s.nrex = nrex;
			HOT.n_ext_left += nlex; HOT.n_ext_right += nrex;
			if (seedmms == 0 && (nlex > 0 || nrex > 0)) {
				const bool m2 = PRM.paired && HOT.pe.cur == 1;
				Work::ExtRange* range = m2 ? (fw ? WK.ex_fw2 : WK.ex_rc2) : (fw ? WK.ex_fw : WK.ex_rc);
				uint32_t& nr = m2 ? (fw ? HOT.pe.n_ex_fw2 : HOT.pe.n_ex_rc2) : (fw ? HOT.n_ex_fw : HOT.n_ex_rc);
				if (nr < (uint32_t)(kMaxRanges * 2)) {
					range[nr].off = rdoff - (fw ? nlex : nrex);
					range[nr].len = seedlen + nlex + nrex;
					range[nr].sz = (uint32_t)sz;
					nr++;
				} else ovf(10);
			}
		}
		}
		nelt_out = nelt;
		HOT.t_phase[17] += now() - th_;       // profile: range collection incl. seed-hit extension that was not pre-computed
		// satpos.sort()
		// (operator< is a total order over the ranges of one read, so any sorting algorithm gives the reference's order;
		// -N 1 can produce thousands of ranges, hence the gapped passes before the final insertion pass)
		const uint32_t gaps[9] = {1750u, 701u, 301u, 132u, 57u, 23u, 10u, 4u, 1u};
		for (int gi = 0; gi < 9; gi++) {
			const uint32_t gap = gaps[gi];
			if (gap >= HOT.n_satpos2) continue;
			for (uint32_t i = gap; i < HOT.n_satpos2; i++) {
				if (!satpos_less(WK.satpos2[i], WK.satpos2[i - gap])) continue;
				const SatPos v = WK.satpos2[i];
				uint32_t j = i;
				while (j >= gap && satpos_less(v, WK.satpos2[j - gap])) { WK.satpos2[j] = WK.satpos2[j - gap]; j -= gap; }
				WK.satpos2[j] = v;
			}
		}
		if (PRM.det_seeds) {
			// prioritizeSATupsIdxs (aligner_sw_driver.cpp:741-866): every range in sorted order until maxelt elements are in
			uint64_t added = 0;
			for (uint32_t j = 0; j < HOT.n_satpos2 && added < maxelt; j++) {
				if (HOT.n_satpos >= (uint32_t)kMaxSatpos) { ovf(11); break; }
				BT2_G SatPos& s = WK.satpos[HOT.n_satpos++];
				s = WK.satpos2[j];
				r1n_init_seq(s.rnd, s.size);
				added += s.size;
			}
			nelt_out = added;
			HOT.n_satpos_full = HOT.n_resolved = HOT.n_satpos;
			return;
		}
		uint64_t nelt_added = 0;
		// 1. the smalls, whole
		for (uint64_t j = 0; j < nsmall && nelt_added < maxelt; j++) {
			if (HOT.n_satpos >= (uint32_t)kMaxSatpos) { ovf(12); break; }
			BT2_G SatPos& s = WK.satpos[HOT.n_satpos++];
			s = WK.satpos2[j];
			r1n_init(s.rnd, s.size, PRM.all_hits != 0);
			nelt_added += s.size;
		}
		HOT.n_satpos_full = HOT.n_resolved = HOT.n_satpos;
		if (nelt_added >= maxelt || nsmall == HOT.n_satpos2) { nelt_out = nelt_added; return; }
		// 2. the non-smalls: RowSampler::init(satpos2_, nsmall, size, lensq=true, szsq=true)
		const uint32_t sai = (uint32_t)nsmall, saf = HOT.n_satpos2;
		HOT.n_masses = saf - sai;
		// With at most 64 candidate ranges (always, for -N 0) the whole sampler runs out of registers (sample_rows_fast)
		const bool fast = HOT.n_masses <= 64u && maxelt + 2u <= 64u * (uint64_t)kSampTabRegs && !big_range;
		HOT.samp_sai = sai; HOT.samp_lanes = HOT.n_masses <= 64u ? 1u : 0u;
		const uint64_t ts_ = now();
		if (fast) {
			nelt_added = Plat::uni(sample_rows_fast(sai, HOT.n_masses, maxelt, nelt, nelt_added));
		} else {
		HOT.mass = 0.0;
		for (uint32_t i = sai; i < saf; i++) {
			WK.masses[i - sai] = samp_mass(WK.satpos2[i].nlex, WK.satpos2[i].nrex, WK.satpos2[i].size);
			WK.elim[i - sai] = 0;
			HOT.mass += WK.masses[i - sai];
		}
		for (uint32_t j = 0; j < HOT.n_satpos2; j++) r1n_reset(WK.rands2[j]);
		while (nelt_added < maxelt && nelt_added < nelt) {
			// RowSampler::next, more ranges than the on-chip sampler holds (-N 1): everything through the arena
			const double rd = (double)(ST.rnd.nextFloat() * HOT.mass);
			uint32_t pick = 0xffffffffu;
			double mass_sofar = 0.0;
			uint32_t last_unelim = 0xffffffffu;
			for (uint32_t i = 0; i < HOT.n_masses; i++) {
				if (!WK.elim[i]) {
					last_unelim = i;
					mass_sofar += WK.masses[i];
					if (rd < mass_sofar) { pick = i; break; }
				}
			}
			if (pick == 0xffffffffu) pick = last_unelim;
			const uint32_t ri = pick + sai;
			BT2_G R1N& r2 = WK.rands2[ri];
			if (!r2.inited) r1n_init(r2, WK.satpos2[ri].size, PRM.all_hits != 0);
			const uint32_t r = r1n_next(r2);
			if (r1n_done(r2)) { WK.elim[ri - sai] = 1; HOT.mass -= WK.masses[ri - sai]; }
			if (HOT.n_satpos >= (uint32_t)kMaxSatpos) { ovf(13); break; }
			BT2_G SampRow* const sr = &WK.srows[HOT.n_satpos - HOT.n_satpos_full];
			HOT.n_satpos++;
			gst(&sr->topf, (uint64_t)(WK.satpos2[ri].topf + r)); gst(&sr->src, ri); gst(&sr->done, 0u);
			nelt_added++;
		}
		}
		HOT.t_phase[18] += now() - ts_;       // profile: the row-sampling loop
		nelt_out = nelt_added;
	}

	// seenDiags1_: plain union-of-intervals semantics (EIvalMergeListBinned, ival_list.h:210)
	BT2_HD bool diag_present(int32_t ref, int64_t off, bool fw) const { return diag_present_m(ref, off, fw, 0); }
	BT2_HD void diag_add(int32_t ref, int64_t off, bool fw, int64_t len) { diag_add_m(ref, off, fw, len, 0); }
	// seenDiags1_ / seenDiags2_ share the list; the mate rides in bit 1 of the orientation
	BT2_HD bool diag_present_m(int32_t ref, int64_t off, bool fw, int mate) const {
		const int orient = (fw ? 1 : 0) | (mate << 1);
		return Plat::diag_find(WK.diags, HOT.n_diags, ref, off, orient);
	}
	BT2_HD void diag_add_m(int32_t ref, int64_t off, bool fw, int64_t len, int mate) {
		if (HOT.n_diags >= (uint32_t)kMaxDiags) { ovf(14); return; }
		BT2_G DiagIval& d = WK.diags[HOT.n_diags++];
		d.ref = ref; d.off = off; d.orient = (fw ? 1 : 0) | (mate << 1); d.len = len;
	}

	// RedundantAlns cell enumeration (aligner_result.cpp:929-1032): per read row, the half-open
	// column range [left,right) the alignment occupies, edits taken WK.r.t. the upstream end.
	struct RowIt {
		const BT2_G AlnRes& r;
		uint32_t n, nedidx, i, end;
		int64_t left;
		bool fw;
		BT2_HD explicit RowIt(const BT2_G AlnRes& r_) : r(r_) {
			fw = r.fw != 0; n = r.nned; nedidx = 0;
			i = fw ? r.trim5p : r.trim3p;           // trimmedLeft(true)
			end = i + r.rdextent;                   // readExtentRows()
			left = r.refoff;
		}
		// edits WK.r.t. the upstream end: for rc alignments positions are inverted (invertPoss) and the order reversed
		BT2_HD uint32_t epos(uint32_t k) const {
			if (fw) return r.ned[k].pos;
			const Edit& e = r.ned[n - 1 - k];
			return (uint32_t)(r.rdextent - e.pos - (e.type == EDIT_READ_GAP ? 0 : 1));
		}
		BT2_HD int etype(uint32_t k) const { return fw ? r.ned[k].type : r.ned[n - 1 - k].type; }
		BT2_HD bool done() const { return i >= end; }
		BT2_HD void next(int64_t& l, int64_t& rgt) {   // range of row i, then step to row i+1
			int64_t diff = 1, right = left + 1;
			while (nedidx < n && epos(nedidx) == i) { if (etype(nedidx) == EDIT_REF_GAP) diff = 0; nedidx++; }
			if (i < end - 1) {
				uint32_t nn = nedidx;
				while (nn < n && epos(nn) == i + 1) { if (etype(nn) == EDIT_READ_GAP) right++; nn++; }
			}
			l = left; rgt = right;
			left = right + diff - 1;
			i++;
		}
	};

	// conservative bounds on (column - row) over the cells of r
	BT2_HD void diag_bounds(const BT2_G AlnRes& r, int64_t& dmin, int64_t& dmax) const {
		const int64_t d0 = r.refoff - (int64_t)(r.fw ? r.trim5p : r.trim3p);
		int64_t nrd = 0, nrf = 0;
		for (uint32_t k = 0; k < r.nned; k++) { const int t = r.ned[k].type; if (t == EDIT_READ_GAP) nrd++; else if (t == EDIT_REF_GAP) nrf++; }
		dmin = d0 - nrf - 1; dmax = d0 + nrd + 1;
	}

	// RedundantAlns::overlap against every alignment reported so far (they are all kept in WK.alns)
	BT2_HDN bool red_overlap(const BT2_G AlnRes& r_) const {
		const BT2_G AlnRes& r = *Plat::uni_ptr(&r_);
		const uint32_t nst = HOT.n_alns < (uint32_t)kMaxAlns ? HOT.n_alns : (uint32_t)kMaxAlns;
		if (nst == 0) return false;
		int64_t dmin, dmax;
		diag_bounds(r, dmin, dmax);
		const int32_t r_refid = r.refid; const bool r_fw = r.fw != 0;
		// every lane checks its own stored alignment (-k 20 on a repeat keeps dozens)
		for (uint32_t a0 = 0; a0 < nst; a0 += Plat::n_lanes()) {
			const uint32_t a = a0 + Plat::lane_id();
			bool hit = false;
			if (a < nst) {
				const BT2_G AlnRes& o = WK.alns[a];
				// alignments whose diagonal ranges are disjoint share no cell
				if (o.refid == r_refid && (o.fw != 0) == r_fw && !(dmin > WK.red_dmax[a] || WK.red_dmin[a] > dmax)) {
					RowIt ia(o), ir(r);
					int64_t l1 = 0, r1 = 0, l2 = 0, r2 = 0;
					while (!hit && !ia.done() && !ir.done()) {
						if (ia.i < ir.i) { ia.next(l2, r2); continue; }
						if (ir.i < ia.i) { ir.next(l1, r1); continue; }
						ia.next(l2, r2); ir.next(l1, r1);
						if (l1 < r2 && l2 < r1) hit = true;
					}
				}
			}
			if (Plat::any(hit)) return true;
		}
		return false;
	}
	// RedundantAlns::add: the cells are re-derived from WK.alns[k] on demand; only the prefilter bounds are kept
	BT2_HDN void red_add(const BT2_G AlnRes& r_) {
		const BT2_G AlnRes& r = *Plat::uni_ptr(&r_);
		if (HOT.n_alns >= (uint32_t)kMaxAlns) return;      // sink_report flags the overflow
		diag_bounds(r, WK.red_dmin[HOT.n_alns], WK.red_dmax[HOT.n_alns]);
	}

	// =================================================================================
	// D. sink (AlnSinkWrap::report, ReportingState::foundUnpaired; aln_sink.cpp:103-130,1395-1445)
	// =================================================================================
	BT2_HD bool sink_report(const BT2_G AlnRes& r) {
		if (HOT.n_alns < (uint32_t)kMaxAlns) Plat::copy_aln(WK.alns[HOT.n_alns], r); else ovf(15);
		HOT.n_alns++;
		if (!HOT.done_unpair1) {
			// ReportingState::areDone
			if (PRM.mhits <= 0 && !PRM.all_hits && HOT.n_alns >= (uint32_t)PRM.khits) { HOT.done_unpair1 = 1; HOT.exit_k = 1; }
			else if (PRM.mhits > 0 && HOT.n_alns > (uint32_t)PRM.mhits) { HOT.done_unpair1 = 1; HOT.exit_m = 1; }
		}
		const int64_t score = r.score;
		if (score > HOT.best_unp1) { HOT.best2_unp1 = HOT.best_unp1; HOT.best_unp1 = score; }
		else if (score > HOT.best2_unp1) HOT.best2_unp1 = score;
		return HOT.done_unpair1 != 0;
	}

	// =================================================================================
	// E. DP: reference window, fill, gather, backtrace
	// =================================================================================
	// SwAligner::initRef (aligner_sw.cpp:155-271): masks for [rect.refl, rect.refr+1], overhang = N
	BT2_HD void fetch_ref_window(uint64_t tidx, int64_t rfi, uint32_t count) {
		// inside the fragment the seed hit was resolved in (nearly always): a contiguous piece of the joined text
		const int64_t rel = rfi - (int64_t)HOT.frag_toff;
		if (tidx == HOT.frag_tidx && rel >= 0 && (uint64_t)rel + count <= HOT.frag_len) {
			Plat::fetch_ref_joined(IX.ref, HOT.frag_jlo + (uint64_t)rel, count);
#ifdef BT2G_CHECK_REF_JOINED
			// test builds: the joined-text form must give what the record search gives
			{ uint8_t a[kMaxColsWide + 8]; memcpy(a, Plat::rf(), count); Plat::fetch_ref(IX.ref, WK, tidx, rfi, count); static unsigned long n_ = 0; n_++;
			  if (memcmp(a, Plat::rf(), count)) { fprintf(stderr, "fetch_ref_joined mismatch (tidx %llu rfi %lld count %u)\n", (unsigned long long)tidx, (long long)rfi, count); abort(); }
			  if ((n_ & (n_ - 1)) == 0) fprintf(stderr, "fetch_ref_joined checked %lu windows\n", n_); }
#endif
		} else Plat::fetch_ref(IX.ref, WK, tidx, rfi, count);
	}

	// packed cell (H | E<<8 | F<<16)
	BT2_HD uint32_t cell_get(uint32_t R, uint32_t i, uint32_t j) const { return ST.dp.mat[dp_cell(R, i, j)]; }

	// masks_ helpers (aligner_swsse.h:255-330,418-490)
	BT2_HD uint16_t& mask_at(uint32_t row, uint32_t col, uint32_t cols) { return ST.dp.masks[(uint64_t)row * cols + col]; }

	// gatherCellsNucleotidesEnd2EndSseU8 (aligner_swsse_ee_u8.cpp:1176-1208) + btncand_.sort()
	BT2_HDN void gather_cells(bool fw_, uint32_t rows_, uint32_t cols_, int64_t minsc_dp_, int mode_, uint32_t lastsolcol_) {
		// (arguments of a real call are lane-varying to the compiler: say that they are not, see DevPlat::dp_fill_ee)
		const bool fw = Plat::uni((int)fw_) != 0;
		const uint32_t rows = Plat::uni(rows_), cols = Plat::uni(cols_), lastsolcol = Plat::uni(lastsolcol_);
		const int64_t minsc_dp = Plat::uni(minsc_dp_);
		const int mode = Plat::uni(mode_);
		const uint32_t R = dp_R(rows);
		HOT.n_cands = 0; HOT.cural = 0; HOT.n_cdone = 0;
		const uint64_t tl_ = now();
		uint32_t nc;
		if (mode != 2) {
			if (mode != 0) Plat::load_last_row(ST.dp.mat, R, rows, cols, true);      // (mode 1, 16-bit end to end; the 8-bit fill leaves the last row in HOT.lastrow itself)
			HOT.t_phase[15] += now() - tl_;
			// btncand_.sort(): score desc, (row desc,) col desc (DpBtCandidate::operator<)
			nc = Plat::gather_sort(cand_list(), (uint32_t)kMaxCands, rows, cols, minsc_dp);
		} else {
			// gatherCellsNucleotidesLocalSseU8/I16 (aligner_swsse_loc_u8.cpp:1389-1496): every cell with score >= ST.minsc, at or
			// below the first row that can reach ST.minsc, that is a match whose diagonal successor is not; columns <= lastsolcol
			const int64_t bonus = PRM.match_bonus;
			const uint32_t minrow = (uint32_t)(((minsc_dp + bonus - 1) / bonus) - 1);
			nc = Plat::gather_local(ST.dp.mat, cand_list(), (uint32_t)kMaxCands, fw, R, rows, lastsolcol + 1, minsc_dp, minrow, WK.cand_hist);
		}
		if (nc > (uint32_t)kMaxCands) { ovf(16); HOT.n_cands = kMaxCands; } else HOT.n_cands = nc;
		HOT.t_phase[13] += HOT.n_cands;      // profile: candidate cells
		if (HOT.n_cands > 0) {      // SSEMatrix::initMasks
			const uint64_t tz_ = now();
			if (mode != 1) {
				// pred format: a new epoch invalidates every mask word of earlier DPs; the plane is only cleared when the tag wraps
				uint32_t e = Plat::uni(*ST.dp.epoch) + 1;
				if (e > kEpochMax) { Plat::zero_u32(ST.dp.pmask, ST.dp.pmask_words); e = 1; }
				Plat::set_epoch(ST.dp.epoch, e);
				Plat::rt_begin(ST.dp, rows, mode == 0);      // (device: the marks of a band matrix live on chip when the launch has room for them)
			} else Plat::zero_masks(ST.dp.masks, rows * cols);
			HOT.t_phase[16] += now() - tz_;
		}
	}

	BT2_HD static int mask2chr(int m) {
		// mask2dna for single-bit masks and N (alphabet.cpp:71); IUPAC multi-bit masks cannot occur
		// because the index only stores A/C/G/T and N
		return (m == 1 || m == 2 || m == 4 || m == 8) ? (int)code2chr(__builtin_ctz((unsigned)m)) : 'N';
	}

	// AlnRes::setShape (aligner_result.cpp:72-122) -- edits arrive WK.r.t. DP rows (upstream end)
	BT2_HD void set_shape(BT2_G AlnRes& r, int32_t id, int64_t off, int64_t reflen, bool fw, uint32_t rdlen, uint32_t trim5p, uint32_t trim3p) {
		r.refid = id; r.refoff = off; r.reflen = reflen; r.fw = fw ? 1 : 0; r.rdlen = (uint16_t)rdlen;
		r.trim5p = (uint16_t)trim5p; r.trim3p = (uint16_t)trim3p;
		const uint32_t trim_beg = fw ? trim5p : trim3p;
		if (trim_beg > 0) for (uint32_t i = 0; i < r.nned; i++) r.ned[i].pos = (uint16_t)(r.ned[i].pos - trim_beg);
		r.rdextent = (uint16_t)(rdlen - trim5p - trim3p);
		int rf = r.rdextent;
		for (uint32_t i = 0; i < r.nned; i++) {
			if (r.ned[i].type == EDIT_REF_GAP) rf--;
			if (r.ned[i].type == EDIT_READ_GAP) rf++;
		}
		r.rfextent = (uint16_t)rf;
	}

	// Edit::invertPoss over the whole list (edit.cpp:50-88): reverse order, pos -> sz - pos - (readgap?0:1)
	BT2_HD void invert_edits(BT2_G AlnRes& r) {
		const uint32_t n = r.nned, sz = r.rdextent;
		for (uint32_t i = 0; i < n / 2; i++) { const Edit t = r.ned[i]; r.ned[i] = r.ned[n - 1 - i]; r.ned[n - 1 - i] = t; }
		for (uint32_t i = 0; i < n; i++) r.ned[i].pos = (uint16_t)(sz - r.ned[i].pos - (r.ned[i].type == EDIT_READ_GAP ? 0 : 1));
	}

	// AlnRes::clipOutside / clipLeft / clipRight with soft clipping (aligner_result.cpp:208-303); Edit::clipLo / clipHi
	// (edit.cpp:438-471).  Clipping is counted in read characters: reference gaps inside the overhang widen it.
	BT2_HD void clip_edits_lo(BT2_G AlnRes& r, uint32_t amt) {
		uint32_t nrm = 0;
		for (uint32_t i = 0; i < r.nned; i++) { if (r.ned[i].pos < amt) nrm++; else r.ned[i].pos = (uint16_t)(r.ned[i].pos - amt); }
		for (uint32_t i = nrm; i < r.nned; i++) r.ned[i - nrm] = r.ned[i];
		r.nned = (uint16_t)(r.nned - nrm);
	}
	BT2_HD void clip_edits_hi(BT2_G AlnRes& r, uint32_t len, uint32_t amt) {
		const uint32_t mx = len - amt;
		while (r.nned > 0) {
			const Edit& e = r.ned[r.nned - 1];
			if (e.pos > mx || (e.pos == mx && e.type != EDIT_READ_GAP)) r.nned--; else break;
		}
	}
	BT2_HD uint32_t clip_read_chars(BT2_G AlnRes& r, bool from_left, uint32_t rf_amt) {
		// walk the edits from the clipped end (positions WK.r.t. that end of the Watson-oriented read)
		const bool inv = from_left ? !r.fw : (r.fw != 0);
		uint32_t rf_i = rf_amt;
		if (inv) invert_edits(r);
		for (uint32_t i = 0; i < r.nned; i++) { if (r.ned[i].pos > rf_i) break; if (r.ned[i].type == EDIT_REF_GAP) rf_i++; }
		if (inv) invert_edits(r);
		return rf_i < r.rdextent ? rf_i : r.rdextent;
	}
	BT2_HDN void clip_outside(BT2_G AlnRes& r_, int64_t refi_, int64_t reff_) {
		BT2_G AlnRes& r = *Plat::uni_ptr(&r_);
		const int64_t refi = Plat::uni(refi_), reff = Plat::uni(reff_);
		if (r.refoff < refi) {
			uint32_t rf_amt = (uint32_t)(refi - r.refoff);
			if (rf_amt > r.rfextent) rf_amt = r.rfextent;
			const uint32_t rd_amt = clip_read_chars(r, true, rf_amt);
			if (r.fw) { r.trim5p = (uint16_t)(r.trim5p + rd_amt); clip_edits_lo(r, rd_amt); }
			else { r.trim3p = (uint16_t)(r.trim3p + rd_amt); clip_edits_hi(r, r.rdextent, rd_amt); }
			r.rdextent = (uint16_t)(r.rdextent - rd_amt); r.rfextent = (uint16_t)(r.rfextent - rf_amt);
			r.refoff += rf_amt;
		}
		const int64_t right = r.refoff + (int64_t)r.rfextent;
		if (right > reff) {
			uint32_t rf_amt = (uint32_t)(right - reff);
			if (rf_amt > r.rfextent) rf_amt = r.rfextent;
			const uint32_t rd_amt = clip_read_chars(r, false, rf_amt);
			if (r.fw) { r.trim3p = (uint16_t)(r.trim3p + rd_amt); clip_edits_hi(r, r.rdextent, rd_amt); }
			else { r.trim5p = (uint16_t)(r.trim5p + rd_amt); clip_edits_lo(r, rd_amt); }
			r.rdextent = (uint16_t)(r.rdextent - rd_amt); r.rfextent = (uint16_t)(r.rfextent - rf_amt);
		}
	}

	// SwAligner::nextAlignment (aligner_sw.cpp:737-1146) with the backtrace (backtraceNucleotidesEnd2EndSseU8, aligner_swsse_ee_u8.cpp:1283-1877,
	// and its 16-bit / local counterparts) inside its candidate loop.  Most candidates of a window fail within a few cells -- they run
	// into the cells an earlier alignment of the same window went through -- so what a backtrace needs before its first step (scoring
	// and rectangle constants in scalar registers, the read / qualities / reference window in lane registers, the geometry of the stored
	// band) is set up ONCE per call, not once per candidate.  fw = orientation aligned.
	// MODE: cell format -- 0 = e2e 8-bit (bias 0xff), 1 = e2e 16-bit (bias 0x7fff), 2 = local (16-bit fields holding plain scores, floor 0)
	template <int MODE>
	BT2_HDN bool next_alignment_m(bool fw_, uint32_t rows_, uint32_t cols_, uint32_t rect_triml, uint32_t rect_corel, uint32_t rect_corer, int64_t rect_refl_,
	                              uint64_t tidx_, int64_t tlen_, bool sse16_, BT2_G AlnRes& res_) {
		BT2_G AlnRes& res = *Plat::uni_ptr(&res_);
		const uint64_t tidx = Plat::uni(tidx_); const int64_t tlen = Plat::uni(tlen_); const bool sse16 = Plat::uni((int)sse16_) != 0;
		if (HOT.cural == HOT.n_cands) return false;
		// Everything below is wave-uniform; Plat::uni() tells the compiler so (scalar registers, scalar ALU).
		const bool fw = Plat::uni((int)fw_) != 0;
		constexpr bool wide = MODE == 1, local = MODE == 2;
		constexpr bool pred = MODE != 1;                 // 8-bit end-to-end and local: one byte of predecessor bits per cell (PB_*), tile = kPredTile steps along the direction of travel
		constexpr uint32_t tile_len = pred ? kPredTile : kBtTile;
		constexpr uint32_t kWalkCap = local ? (uint32_t)kMaxWalkEdits : (uint32_t)kMaxEdits;      // edits a walk may collect (hot_tail_bytes sizes ned[] the same way)
		const uint32_t rows = Plat::uni(rows_), cols = Plat::uni(cols_);
		struct { int gapbar, rdgapo, rdgape, rfgapo, rfgape, match_bonus, mm_type, mm_max, mm_min, n_pen; } S;
		S.gapbar = Plat::uni(PRM.gapbar); S.rdgapo = Plat::uni(PRM.rdgapo); S.rdgape = Plat::uni(PRM.rdgape);
		S.rfgapo = Plat::uni(PRM.rfgapo); S.rfgape = Plat::uni(PRM.rfgape); S.match_bonus = Plat::uni(PRM.match_bonus);
		S.mm_type = Plat::uni(PRM.mm_type); S.mm_max = Plat::uni(PRM.mm_max); S.mm_min = Plat::uni(PRM.mm_min); S.n_pen = Plat::uni(PRM.n_pen);
		// (the rectangle's fields arrive by value: a reference to the caller's DPRect would force it into scratch memory)
		const int r_triml = (int)Plat::uni(rect_triml), r_corel = (int)Plat::uni(rect_corel), r_corer = (int)Plat::uni(rect_corer);
		const int64_t rect_refl = Plat::uni(rect_refl_);
		const uint32_t R = dp_R(rows);
		// `this` lives in private memory: read what the loop needs once, into scalar registers
		DpScratch dpl;
		dpl.mat = Plat::uni_ptr(ST.dp.mat); dpl.masks = Plat::uni_ptr(ST.dp.masks); dpl.pmask = Plat::uni_ptr(ST.dp.pmask); dpl.epoch = Plat::uni_ptr(ST.dp.epoch); dpl.pmask_words = 0;
		const uint32_t epoch = pred ? Plat::uni(*dpl.epoch) : 0u;
		const int32_t band_lo = pred ? (int32_t)Plat::uni(dpl.epoch[1]) : 0;       // geometry of the band the fill stored (pred_idx)
		const uint32_t band_w = pred ? Plat::uni(dpl.epoch[2]) : 0u;
		BtCand* const cands = Plat::uni_ptr(cand_list());
		struct Prof {      // profile counters stay in registers until the function returns
			uint32_t steps, tiles; uint64_t tile_t; uint32_t scalar_steps;
			BT2_HD ~Prof() { ST.pf_steps += steps; ST.pf_tiles += tiles; ST.pf_tile_t += tile_t; HOT.t_bt[4] += scalar_steps; }
		} prof{0, 0, 0, 0};
		// read, qualities and reference window as per-lane registers (4 bytes per lane per register): the step
		// loop then reads them with v_readlane instead of going to LDS
		constexpr uint32_t kSqRegs = ((uint32_t)kMaxLen + 255u) / 256u, kRfRegs = kSqRegs + 1u, kRfWin = 256u * kRfRegs;      // 256 bytes per register
		typename Plat::LaneReg sqw[kSqRegs], qlw[kSqRegs], rfw[kRfRegs];
		// The backtrace's branch stack (btnstack_, DpNucFrame) and the per-cell H/E/F choice masks (SSEMatrix::masks_ bits 1-12) are DEAD STATE
		// in the reference: every cell a walk visits gets reportedThrough set in the same visit that stores its choice mask
		// (aligner_swsse_ee_u8.cpp:1331-1338 tests reportedThrough first, :1556 sets it for every visited cell), so (a) a choice mask is never read
		// again -- any later visit stops at "reportedThru" before it looks -- and (b) a popped frame resumes in a cell this very walk has marked,
		// fails at once and pops the next (:1560-1586): a walk that meets a marked cell fails, after one more loop iteration per frame on the
		// stack (they count as backtrace cells, met.btcell).  What a walk needs of the matrix is the predecessor bits and ONE bit per cell.
		for (uint32_t k = 0; k < kSqRegs; k++) { sqw[k] = Plat::lanes_load(HOT.seq, kMaxLen, k * 64); qlw[k] = Plat::lanes_load(HOT.qual, kMaxLen, k * 64); }
		// the walk moves left from its start column by at most rows + gaps columns: one register more than the read needs (768 columns in the general class) covers every window of an
		// unpaired read from column 0 (rf_c0 = 0); only a candidate past column 767 of a wide opposite-mate window needs them re-based
		uint32_t rf_c0_ = 0;
		for (uint32_t k = 0; k < kRfRegs; k++) rfw[k] = Plat::lanes_load(Plat::rf(), ST.max_cols + 8u, k * 64);
		auto byte_of = [](typename Plat::LaneReg* arr, uint32_t nreg, uint32_t idx) -> int {
			const uint32_t word = idx >> 2;
			uint32_t v = Plat::lane(arr[0], word & 63);
#pragma unroll
			for (uint32_t k = 1; k < kRfRegs; k++) if (k < nreg && (word >> 6) == k) v = Plat::lane(arr[k], word & 63);
			return (int)((v >> ((idx & 3) * 8)) & 0xff);
		};
		// one backtrace from cell (row, col), whose tile the caller fetched
		auto walk = [&](uint32_t row, uint32_t col, typename Plat::LaneReg tile, typename Plat::LaneReg tile_hi) __attribute__((always_inline)) -> bool {
			row = Plat::uni(row); col = Plat::uni(col);
			const uint32_t rf_c0 = Plat::uni(rf_c0_);
			uint32_t td = 0;     // td = steps taken along the tile
			uint32_t tdir = 0, ndir = 0;      // pred format: direction of the tile in hand / of the next fetch (0 diagonal, 1 left along the row, 2 up the column)
			const uint32_t rdlen = rows;   // end-to-end: one DP row per read character
			int olap = 0;        // the path touches a core diagonal of the untrimmed rectangle (:1764-1795)
			auto in_core = [&](uint32_t r_, uint32_t c_) -> int {
				const int diagi = (int)c_ - (int)r_ + r_triml;          // rows, columns and trims are all < 2^16
				return (int)(diagi >= r_corel) & (int)(diagi <= r_corer);     // corel >= 0, so diagi >= 0 is implied
			};
			uint32_t nstack = 0, ncells = 0, nned = 0;
			int32_t score = 0, ns = 0;
			const uint32_t orig_col = col;
			uint32_t gaps = 0, read_gaps = 0, ref_gaps = 0;
			const uint32_t trim_end = rows - row - 1;
			uint32_t trim_beg = 0;
			int ct = 0;      // 0=H 1=E 2=F (SSEMatrix::H/E/F order irrelevant here)
			Edit* ned = Plat::ned();
			const int offsetsc = local ? 0 : (wide ? -0x7fff : -0xff);
			HOT.n_bt_attempts++;
			while ((int)row >= 0) {
				// (the loop's own state is wave-uniform; said once per iteration, so that nothing lane-varying that feeds one of these variables on
				// some path can turn the whole loop into vector code under exec masks)
				row = Plat::uni(row); col = Plat::uni(col); td = Plat::uni(td); tdir = Plat::uni(tdir); ndir = Plat::uni(ndir);
				ct = Plat::uni(ct); nned = Plat::uni(nned); ncells = Plat::uni(ncells); nstack = Plat::uni(nstack); score = Plat::uni(score); ns = Plat::uni(ns);
				gaps = Plat::uni(gaps); olap = Plat::uni(olap);
				if (pred && ct != 0 && tdir == (uint32_t)ct && td < tile_len && row > 0) {
					// inside a gap: the cells that can only EXTEND it (unvisited, E consistent with E-left alone / F with F-up alone) are walked
					// by all lanes at once -- same marks, same edits, same counters as the step-by-step loop below.  (The candidates next to an
					// alignment's end column each walk a gap of growing length back to its path; the cell that opens the gap and whatever
					// follows go through the scalar step.)
					const uint32_t room_c = ncells < (uint32_t)(kMaxLen + 64) ? (uint32_t)(kMaxLen + 64) - ncells : 0u;
					const uint32_t room_e = nned + 2 < kWalkCap ? kWalkCap - 2 - nned : 0u;
					const uint32_t room = room_c < room_e ? room_c : room_e;
					uint32_t core = 0;
					const uint32_t L = Plat::uni(Plat::bt_gap_run(dpl, band_lo, band_w, epoch, tile, tile_hi, td, row, col, ct == 1, fw, rdlen, room,
					                                              nned, r_triml, r_corel, r_corer, core));
					if (L > 0) {
						olap |= (int)(Plat::uni(core) != 0); ncells += L; prof.steps += L; nned += L; gaps += L; td += L;
						if (ct == 1) { read_gaps += L; col -= L; score -= (int32_t)L * S.rdgape; }
						else { ref_gaps += L; row -= L; score -= (int32_t)L * S.rfgape; }
						continue;
					}
				}
				if (pred && ct == 0 && tdir == 0 && td < tile_len && row > 0) {
					// a run of plain diagonal steps (unvisited cells whose only consistent predecessor is the diagonal one) is walked
					// by all lanes at once: same marks, same edits, same counters as the step-by-step loop below
					const uint32_t room_c = ncells < (uint32_t)(kMaxLen + 64) ? (uint32_t)(kMaxLen + 64) - ncells : 0u;
					const uint32_t room_e = nned + 2 < kWalkCap ? kWalkCap - 2 - nned : 0u;
					const uint32_t room = room_c < room_e ? room_c : room_e;
					typename Plat::LaneReg inf;
					uint64_t mm;
					const uint32_t L = Plat::uni(Plat::bt_diag_run(dpl, band_lo, band_w, epoch, tile, tile_hi, td, row, col, fw, rdlen, room, inf, mm));
					if (L > 0) {
						olap |= in_core(row, col); ncells += L; prof.steps += L;
						if (local) score += (int32_t)(L - (uint32_t)__builtin_popcountll(mm)) * S.match_bonus;      // (end to end a match scores 0)
						while (mm) {
							const uint32_t d = (uint32_t)__builtin_ctzll(mm);
							mm &= mm - 1;
							const uint32_t v = Plat::lane(inf, d);
							const int e_readc = (int)((v >> 4) & 7), e_refm = (int)((v >> 8) & 0xff), e_q = (int)((v >> 16) & 0xff);
							Edit& e = ned[nned++];
							e.pos = (uint16_t)(row - (d - td)); e.chr = (uint8_t)mask2chr(e_refm); e.qchr = code2chr(e_readc); e.type = EDIT_MM;
							score -= sc_mm(S, e_readc, e_refm, e_q - 33);
							if (v & 2u) ns++;
						}
						row -= L; col -= L; td += L;
						continue;
					}
				}
				const int readc = fw ? byte_of(sqw, kSqRegs, row) : comp4(byte_of(sqw, kSqRegs, rdlen - 1 - row));
				const int refm = byte_of(rfw, kRfRegs, col - rf_c0);
				const int readq = byte_of(qlw, kSqRegs, fw ? row : rdlen - 1 - row);
				// Flags are ints combined with & and |: the control code is wave-uniform and this keeps it on 32-bit scalar
				// compares/selects instead of 64-bit lane-mask juggling.
				int empty = 0, can_move_thru = 1, branch = 0;
				int cur = 0;   // 0 diag, 1 ref-open (H up), 2 rfgap-extend (F up), 3 read-open (H left), 4 rdgap-extend (E left)
				prof.steps++; prof.scalar_steps++;
				if (td >= tile_len) {
					const uint64_t tt_ = now();
					if (pred) { Plat::bt_tile_pred(dpl, band_lo, band_w, row, col, epoch, ndir, tile, tile_hi); tdir = ndir; }
					else Plat::bt_tile(dpl, R, cols, row, col, wide, tile, tile_hi);
					td = 0; prof.tiles++; prof.tile_t += now() - tt_;
				}
				const uint32_t mk0 = pred ? Plat::lane(tile_hi, td) : (Plat::lane(tile, 48 + td) & 0xffffu);
				if (mk0 & 1) {                    // reportedThrough
					can_move_thru = 0;
				} else if (row > 0) {
					int mask, sel = -1;
					if (pred) {
						// the fill already answered "which predecessors are score-consistent" (PB_* bits, gap barrier folded in)
						const int pb = (int)Plat::lane(tile, td);
						if (ct == 1) {
							mask = (pb >> 3) & 3;
							branch = (int)(mask == 3);
							if (mask != 0) { cur = (mask == 2) ? 4 : 3; sel = 0; }
						} else if (ct == 2) {
							mask = (pb >> 5) & 3;
							branch = (int)(mask == 3);
							if (mask != 0) { cur = (mask == 2) ? 2 : 1; sel = 0; }
						} else {
							const int he = (pb >> 1) & 1, hf = (pb >> 2) & 1;
							mask = (hf & (pb >> 5) & 1) | ((he & (pb >> 3) & 1) << 1) | ((hf & (pb >> 6) & 1) << 2) | ((he & (pb >> 4) & 1) << 3) | ((pb & 1) << 4);
							if (mask != 0) {
								sel = (mask & 16) ? 4 : (mask & 1) ? 0 : (mask & 4) ? 2 : (mask & 2) ? 1 : 3;
								branch = (int)((mask & (mask - 1)) != 0);
								cur = (int)((0x04231u >> (4 * sel)) & 7);
							}
						}
					} else {
					const uint32_t row_from_end = rows - row - 1;
					const int ga = (int)(row >= (uint32_t)S.gapbar) & (int)(row_from_end >= (uint32_t)S.gapbar);     // gaps allowed
					auto cell = [&](uint32_t ln) -> uint64_t { return (uint64_t)Plat::lane(tile, ln) | (wide ? (uint64_t)Plat::lane(tile_hi, ln) << 32 : 0ull); };
					const uint64_t c_cur = cell(td);
					const uint64_t c_up = cell(16 + td);
					const uint64_t c_left = cell(32 + td);
					const uint64_t c_upleft = cell(td + 1);
					auto Hc = [&](uint64_t c) -> int { return local ? (int)(c & 0xffff) : wide ? (int)(int16_t)(uint16_t)(c & 0xffff) : (int)(c & 0xff); };
					auto Ec = [&](uint64_t c) -> int { return local ? (int)((c >> 16) & 0xffff) : wide ? (int)(int16_t)(uint16_t)((c >> 16) & 0xffff) : (int)((c >> 8) & 0xff); };
					auto Fc = [&](uint64_t c) -> int { return local ? (int)((c >> 32) & 0xffff) : wide ? (int)(int16_t)(uint16_t)((c >> 32) & 0xffff) : (int)((c >> 16) & 0xff); };
					auto fl = [&](int v) -> int { return local ? (int)(v > 0) : 1; };    // `> floorsc` of the local kernels (aligner_swsse_loc_u8.cpp:1530-1660)
					const int hasl = (int)(col > 0);
					if (ct == 1) {          // E: came from the left (H-left open = bit 0, E-left extend = bit 1)
						const int sc_cur = Ec(c_cur) + offsetsc, sc_h_left = Hc(c_left) + offsetsc, sc_e_left = Ec(c_left) + offsetsc;
						mask = (fl(sc_h_left) & (int)(sc_h_left - S.rdgapo == sc_cur)) | ((fl(sc_e_left) & (int)(sc_e_left - S.rdgape == sc_cur)) << 1);
						// both -> take the open (cur 3); else the one there is
						branch = (int)(mask == 3);
						if (mask != 0) { cur = (mask == 2) ? 4 : 3; sel = 0; }
					} else if (ct == 2) {   // F: came from above (H-up open = bit 0, F-up extend = bit 1)
						const int sc_cur = Fc(c_cur) + offsetsc, sc_h_up = Hc(c_up) + offsetsc, sc_f_up = Fc(c_up) + offsetsc;
						mask = (fl(sc_h_up) & (int)(sc_h_up - S.rfgapo == sc_cur)) | ((fl(sc_f_up) & (int)(sc_f_up - S.rfgape == sc_cur)) << 1);
						branch = (int)(mask == 3);
						if (mask != 0) { cur = (mask == 2) ? 2 : 1; sel = 0; }
					} else {                // H: bit 0 ref-gap open, 1 read-gap open, 2 ref-gap extend, 3 read-gap extend, 4 diagonal
						const int sc_cur = Hc(c_cur) + offsetsc;
						const int sc_f_up = Fc(c_up) + offsetsc, sc_h_up = Hc(c_up) + offsetsc;
						const int sc_h_left = Hc(c_left) + offsetsc, sc_e_left = Ec(c_left) + offsetsc, sc_h_upleft = Hc(c_upleft) + offsetsc;
						const int sc_diag = sc_score(S, readc, refm, readq - 33);
						mask = (ga & fl(sc_h_up) & (int)(sc_cur == sc_h_up - S.rfgapo))
						     | ((ga & hasl & fl(sc_h_left) & (int)(sc_cur == sc_h_left - S.rdgapo)) << 1)
						     | ((ga & fl(sc_f_up) & (int)(sc_cur == sc_f_up - S.rfgape)) << 2)
						     | ((ga & hasl & fl(sc_e_left) & (int)(sc_cur == sc_e_left - S.rdgape)) << 3)
						     | ((hasl & fl(sc_h_upleft) & (int)(sc_cur == sc_h_upleft + sc_diag)) << 4);
						if (mask != 0) {
							// preference: diagonal, ref-gap open, ref-gap extend, read-gap open, read-gap extend (the only option if there is one)
							sel = (mask & 16) ? 4 : (mask & 1) ? 0 : (mask & 4) ? 2 : (mask & 2) ? 1 : 3;
							branch = (int)((mask & (mask - 1)) != 0);           // more than one option: the reference pushes a frame
							cur = (int)((0x04231u >> (4 * sel)) & 7);          // sel 0,1,2,3,4 -> cur 1,3,2,4,0
						}
					}
					}
					if (sel < 0) empty = 1;      // (canMoveThru = origMask == 0 always holds here: the stored mask equals the computed one, see above)
				}
				if (!(mk0 & 1)) {                // setReportedThrough
					if (pred) Plat::rt_mark(dpl, band_lo, band_w, epoch, row, col);
					else dpl.masks[(uint64_t)row * cols + col] = (uint16_t)1;
				}
				if (!can_move_thru) {
					// every frame on the branch stack resumes in a cell this walk has marked: one more loop iteration each, then the walk fails
					prof.steps += nstack; prof.scalar_steps += nstack;
					return false;
				}
				if (empty || row == 0) {
					olap |= in_core(row, col); ncells++;
					trim_beg = row;
					break;
				}
				if (branch) nstack++;            // (a frame the reference pushes; only its count matters, see above)
				if (ncells >= (uint32_t)(kMaxLen + 64)) { ovf(19); return false; }
				olap |= in_core(row, col); ncells++;
				if (nned + 1 >= kWalkCap) { ovf(20); return false; }
				switch (cur) {
					case 0: {   // diagonal
						const int m = (refm >= 16 || readc > 3) ? -1 : (((1 << readc) & refm) ? 1 : 0);
						ct = 0;
						if (m != 1) {
							Edit& e = ned[nned++];
							e.pos = (uint16_t)row; e.chr = (uint8_t)mask2chr(refm); e.qchr = code2chr(readc); e.type = EDIT_MM;
							score -= sc_mm(S, readc, refm, readq - 33);
						} else {
							score += S.match_bonus;
						}
						if (m == -1) ns++;
						row--; col--;
						if (pred && tdir != 0) { td = tile_len; ndir = 0; } else td++;
						break;
					}
					case 1: case 2: {   // ref gap (move up): open from H / extend from F
						Edit& e = ned[nned++];
						e.pos = (uint16_t)row; e.chr = '-'; e.qchr = code2chr(readc); e.type = EDIT_REF_GAP;
						row--;
						ct = (cur == 1) ? 0 : 2;
						// inside a reference gap (F state: the next move is up again) one tile runs up the column for the whole gap
						if (pred && ct == 2 && tdir == 2) td++; else { td = tile_len; ndir = (pred && ct == 2) ? 2u : 0u; }
						score -= (cur == 1) ? S.rfgapo : S.rfgape;
						gaps++; ref_gaps++;
						break;
					}
					default: {          // read gap (move left): open from H / extend from E
						Edit& e = ned[nned++];
						e.pos = (uint16_t)(row + 1); e.chr = (uint8_t)mask2chr(refm); e.qchr = '-'; e.type = EDIT_READ_GAP;
						col--;
						ct = (cur == 3) ? 0 : 1;
						// inside a read gap (E state: the next move is left again) one tile runs left along the row
						if (pred && ct == 1 && tdir == 1) td++; else { td = tile_len; ndir = (pred && ct == 1) ? 1u : 0u; }
						score -= (cur == 3) ? S.rdgapo : S.rdgape;
						gaps++; read_gaps++;
						break;
					}
				}
			}
			if (!olap) return false;
			const uint64_t tt0_ = now();
			struct TailTimer { uint64_t t0; BT2_HD ~TailTimer() { if (PRM.profile) HOT.t_bt[3] += now() - t0; } } tail_timer_{tt0_};
			{
				const int readc = fw ? byte_of(sqw, kSqRegs, row) : comp4(byte_of(sqw, kSqRegs, rdlen - 1 - row));
				if (col < rf_c0) { ovf(21); return false; }
				const int refm = byte_of(rfw, kRfRegs, col - rf_c0);
				const int m = (refm >= 16 || readc > 3) ? -1 : (((1 << readc) & refm) ? 1 : 0);
				if (m != 1) {
					Edit& e = ned[nned++];
					e.pos = (uint16_t)row; e.chr = (uint8_t)mask2chr(refm); e.qchr = code2chr(readc); e.type = EDIT_MM;
					score -= sc_mm(S, readc, refm, byte_of(qlw, kSqRegs, fw ? row : rdlen - 1 - row) - 33);
				} else score += S.match_bonus;
				if (m == -1) ns++;
			}
			if (ns > RPR.nceil) return false;
			if (kWalkCap > (uint32_t)kMaxEdits && nned > (uint32_t)kMaxEdits) { ovf(20); return false; }      // an alignment with more edits than a result slot holds
			// res.reverse(), while copying the edits out of LDS
			for (uint32_t i = 0; i < nned; i++) res.ned[i] = ned[nned - 1 - i];
			res.nned = (uint16_t)nned;
			res.score = score; res.ns = (int16_t)ns; res.gaps = (int16_t)gaps; res.edits = (int16_t)nned;
			res.bases_aligned = (int16_t)((int)rows - (int)trim_beg - (int)trim_end - (int)nned);
			uint32_t refns = 0;
			for (uint32_t i = col; i <= orig_col; i++) if (Plat::rf()[i] > 15) refns++;
			res.refns = (uint16_t)refns;
			set_shape(res, (int32_t)tidx, (int64_t)col + rect_refl, tlen, fw, rows, fw ? trim_beg : trim_end, fw ? trim_end : trim_beg);
			return true;
	
		};
		bool found = false;
		// the next 64 candidates wait in lane registers (one gather instead of one dependent load per candidate)
		typename Plat::LaneReg cw0, cw1;
		uint32_t cbase = 0xffffff00u;
		// local mode: the candidates of this window that have been tried (btncanddone_: a few dozen of the thousands a 400-bp window can
		// have) as row | col << 16, listed in the arena and mirrored in two lane registers (the first 128): a candidate is tested against
		// all of them at once instead of against every earlier candidate one dependent load at a time
		BT2_G uint32_t* const donel = Plat::uni_ptr(&WK.cand_done[ST.cands_cur == WK.cands2 ? 1 : 0][0]);
		uint32_t ndone = MODE == 2 ? HOT.n_cdone : 0u;
		typename Plat::LaneReg dn0, dn1;
		Plat::lanes_zero(dn0); Plat::lanes_zero(dn1);
		if (MODE == 2 && ndone > 0) { dn0 = Plat::lanes_load_u32(donel, 0, ndone); if (ndone > 64u) dn1 = Plat::lanes_load_u32(donel, 64, ndone); }
		// local mode: which of the 64 candidates in the lanes are dominated by a candidate already tried (bit per lane, kept up to date as
		// candidates are tried): a 400-bp window has thousands of candidate cells and all but a few dozen are dominated -- they are skipped
		// 64 at a time (one ballot) instead of one test per candidate
		typename Plat::LaneReg domv;
		Plat::lanes_zero(domv);
		uint32_t SQ = rows >> 4; if (SQ == 0) SQ = 1;      // "dominated": within SQ rows and columns of a tried candidate (aligner_sw.cpp:754-755,936-960)
		// ---- the candidates that die within a few cells, side by side (round 6; end-to-end 8-bit matrices) ----
		// Once a window has an alignment, nearly every further candidate cell of its last row fails: its backtrace runs into a cell an earlier
		// walk went through (23 of 24 attempts of the headline workload; 4 to 30 cells in, the gap barrier keeps the first ones on the diagonal).
		// A walk draws nothing from the read's RNG and its path is a function of the predecessor bytes alone -- only WHERE IT STOPS depends on the
		// marks the earlier walks left.  So:
		//  (A) every lane follows the path of its own candidate through the predecessor bytes, eight cells per memory round trip (a tile in the
		//      direction of travel per lane), up to the first cell that is marked ALREADY (the walk cannot get past it), the cell that ends the
		//      walk (row 0 / no predecessor), or kBwSteps cells -- and keeps the path as two bit masks: moves that go up a row, moves that go
		//      left a column;
		//  (B) the candidates are then taken in their order: lane i looks at cell i of the candidate's path (its position: two popcounts) -- marks
		//      left by the candidates before it in this very batch included.  First cell marked -> skipped, no draw (aligner_sw.cpp:794-795);
		//      a marked cell further on -> a failed attempt: the cells before it are marked, and it is booked exactly as the step loop books it
		//      (attempt, reseeding draws, cells + one per branch frame).  No marked cell on the path: not a quick failure -- this candidate,
		//      and whatever follows, goes through the step loop below.
		constexpr uint32_t kBwSteps = 32u;
		bool bw_ok = MODE == 0 && HOT.cural > 0u && Plat::marks_batchable();      // (the first candidate of a window is the one that succeeds)
		auto batch_walks = [&]() {
			const uint32_t first = HOT.cural - cbase;
			const uint32_t nv = HOT.n_cands - cbase < 64u ? HOT.n_cands - cbase : 64u;
			typename Plat::LaneReg row, col, ctr, alive, upm, lfm, ncell, brm, elig;
			const int64_t msc = Plat::uni(ST.minsc);
			BT2_FOR_LANES(l) {
				const uint32_t rc_ = LV(cw1);
				const bool e = l >= first && l < nv && (int64_t)(int32_t)LV(cw0) >= msc && (rc_ >> 16) + 1u <= kRfWin;
				LV(elig) = e ? 1u : 0u; LV(alive) = LV(elig);
				LV(row) = rc_ & 0xffffu; LV(col) = rc_ >> 16; LV(ctr) = 0u; LV(upm) = LV(lfm) = 0u; LV(ncell) = 1u; LV(brm) = 0u;
			}
			// candidates from the first one that is not eligible on are the step loop's
			uint32_t limit = nv;
			{ typename Plat::LaneReg ne; BT2_FOR_LANES(l) { LV(ne) = (l >= first && l < nv && !LV(elig)) ? 1u : 0u; }
			  const uint64_t m = Plat::ballot(ne); if (m) limit = (uint32_t)__builtin_ctzll(m); }
			if (limit < first + 2u) { bw_ok = false; return; }
			const typename Plat::LaneReg row0 = row, col0 = col;
			// (A) the paths
			for (uint32_t trip = 0; trip < kBwSteps; trip++) {      // (a trip advances every live lane by at least one cell)
				if (!Plat::ballot(alive)) break;
				typename Plat::LaneReg plo, phi, mk8, intile;
				const typename Plat::LaneReg tdir = ctr;      // H: up the diagonal, E: left along the row, F: up the column
				Plat::pred_tile8(dpl, band_lo, band_w, epoch, row, col, tdir, alive, plo, phi, mk8);
				BT2_FOR_LANES(l) { LV(intile) = LV(alive); }
#pragma unroll
				for (uint32_t i = 0; i < 8u; i++) {
					BT2_FOR_LANES(l) {
						if (LV(intile)) {
							uint32_t r = LV(row), c = LV(col);
							const int ct = (int)LV(ctr), pb = (int)(((i < 4u ? LV(plo) : LV(phi)) >> (8u * (i & 3u))) & 0xffu);
							if ((LV(mk8) >> i) & 1u) { LV(alive) = 0u; LV(intile) = 0u; }      // marked already: the walk stops here at the latest
							else {
								int cur = -1, branch = 0;
								if (r > 0u) {
									if (ct == 1) { const int mask = (pb >> 3) & 3; branch = (int)(mask == 3); if (mask != 0) cur = (mask == 2) ? 4 : 3; }
									else if (ct == 2) { const int mask = (pb >> 5) & 3; branch = (int)(mask == 3); if (mask != 0) cur = (mask == 2) ? 2 : 1; }
									else {
										const int he = (pb >> 1) & 1, hf = (pb >> 2) & 1;
										const int mask = (hf & (pb >> 5) & 1) | ((he & (pb >> 3) & 1) << 1) | ((hf & (pb >> 6) & 1) << 2) | ((he & (pb >> 4) & 1) << 3) | ((pb & 1) << 4);
										if (mask != 0) {
											const int sel = (mask & 16) ? 4 : (mask & 1) ? 0 : (mask & 4) ? 2 : (mask & 2) ? 1 : 3;
											branch = (int)((mask & (mask - 1)) != 0);
											cur = (int)((0x04231u >> (4 * sel)) & 7);
										}
									}
								}
								if (cur < 0) { LV(alive) = 0u; LV(intile) = 0u; }      // the walk ends in this cell (row 0, or no predecessor)
								else {
									const uint32_t n = LV(ncell) - 1u;      // this cell's number = the number of its move
									uint32_t mv;      // 0 diagonal, 1 left, 2 up
									if (cur == 0) { mv = 0u; r--; c--; LV(ctr) = 0u; }
									else if (cur == 1 || cur == 2) { mv = 2u; r--; LV(ctr) = cur == 1 ? 0u : 2u; }
									else { mv = 1u; c--; LV(ctr) = cur == 3 ? 0u : 1u; }
									if (mv != 1u) LV(upm) = LV(upm) | (1u << n);
									if (mv != 2u) LV(lfm) = LV(lfm) | (1u << n);
									if (branch) LV(brm) = LV(brm) | (1u << n);
									LV(row) = r; LV(col) = c; LV(ncell) = n + 2u;
									if (n + 1u == kBwSteps) { LV(alive) = 0u; LV(intile) = 0u; }
									else if (mv != LV(tdir)) LV(intile) = 0u;      // the path leaves the tile's line
								}
							}
						}
					}
				}
			}
			// (B) the candidates in their order, over the marks
			uint32_t last = Plat::uni(ST.rnd.last), lastOff = Plat::uni(ST.rnd.lastOff);
			uint32_t natt = 0, nsteps = 0, k = first;
			for (; k < limit; k++) {
				const uint32_t r0 = Plat::lane(row0, k), c0 = Plat::lane(col0, k), nc = Plat::lane(ncell, k), um = Plat::lane(upm, k), lm = Plat::lane(lfm, k);
				typename Plat::LaneReg cr, cc, cf;
				BT2_FOR_LANES(l) {
					const uint32_t below = l >= 32u ? 0xffffffffu : (1u << l) - 1u;
					LV(cf) = l < nc ? 1u : 0u;
					LV(cr) = r0 - (uint32_t)__builtin_popcount(um & below); LV(cc) = c0 - (uint32_t)__builtin_popcount(lm & below);
				}
				const uint64_t mk = Plat::ballot(Plat::marks_of_cells(dpl, band_lo, band_w, epoch, cr, cc, cf));
				if (mk & 1ull) continue;      // reportedThrough already: no attempt, no draw
				if (!mk) break;               // no marked cell on the path as far as it was followed: the step loop walks this candidate
				const uint32_t t = (uint32_t)__builtin_ctzll(mk);
				BT2_FOR_LANES(l) { LV(cf) = l < t ? 1u : 0u; }
				Plat::mark_cells(dpl, band_lo, band_w, epoch, cr, cc, cf);
				// a failed attempt: reseeding draws as the step loop makes them (8-bit kernels: init(reseed) ... init(reseed + 1)); cells visited,
				// the marked one included, + one more loop iteration per branch frame
				{ Rng g2; g2.last = last; g2.lastOff = lastOff; const uint32_t reseed = g2.nextU32() + 1u; g2.init(reseed + 1u); last = g2.last; lastOff = g2.lastOff; }
#ifdef BT2G_SAMP_STATS
				{ static unsigned long h_[34], n_ = 0; h_[t < 33u ? t : 33u]++; n_++;
				  if ((n_ & (n_ - 1)) == 0) { fprintf(stderr, "WALKSTAT batch-resolved %lu: death index", n_); for (int q = 0; q < 34; q++) fprintf(stderr, " %d:%lu", q, h_[q]); fprintf(stderr, "\n"); } }
#endif
				natt++;
				nsteps += t + 1u + (uint32_t)__builtin_popcountll((uint64_t)Plat::lane(brm, k) & ((1ull << t) - 1ull));
			}
			if (k > first) {
				ST.rnd.last = last; ST.rnd.lastOff = lastOff;
				HOT.n_bt_attempts += natt; prof.steps += nsteps;
				HOT.cural = cbase + k;
			}
			if (k < limit || natt == 0u) bw_ok = false;      // what stopped the batch goes through the step loop first
		};
		while (HOT.cural < HOT.n_cands) {
			BtCand c;
			{
				uint32_t ci = HOT.cural;
				if (ci - cbase >= 64u) {
					cbase = ci; Plat::lanes_load_cands(cands, cbase, HOT.n_cands, cw0, cw1);
					if (MODE == 2) {
						Plat::lanes_zero(domv);
						for (uint32_t k = 0; k < ndone; k++) {
							const uint32_t v_ = k < 64u ? Plat::lane(dn0, k) : k < 128u ? Plat::lane(dn1, k - 64u) : Plat::uni(gld(donel + k));
							Plat::dom_update(domv, cw1, v_, SQ);
						}
					}
				}
				if (MODE == 0 && bw_ok) {
					batch_walks();
					if (HOT.cural != ci) continue;
				}
				if (MODE == 2) {
					// the next candidate that is not dominated -- unless one below the minimum score comes first (the list is sorted by score:
					// that one ends the window, exactly where the one-by-one scan would have met it)
					bool low_first = false;
					const uint32_t nv = HOT.n_cands - cbase < 64u ? HOT.n_cands - cbase : 64u;
					const uint32_t nx = Plat::next_cand(cw0, domv, ci - cbase, nv, ST.minsc, low_first);
					if (low_first) { HOT.cural = HOT.n_cands; break; }
					if (nx >= nv) { HOT.cural = cbase + nv; continue; }      // the rest of this batch is dominated
					ci = cbase + nx; HOT.cural = ci;
				}
				c.score = (int32_t)Plat::lane(cw0, ci - cbase);
				const uint32_t rc_ = Plat::lane(cw1, ci - cbase);
				c.row = (uint16_t)(rc_ & 0xffffu); c.col = (uint16_t)(rc_ >> 16);
			}
			if (c.score < ST.minsc) { HOT.cural = HOT.n_cands; break; }    // sorted by score: every later candidate is filtered too (no RNG draw involved)
			typename Plat::LaneReg tile, tile_hi;
			{
				const uint64_t tt_ = now();      // also the first tile of the backtrace
				if (pred) Plat::bt_tile_pred(dpl, band_lo, band_w, c.row, c.col, epoch, 0u, tile, tile_hi); else Plat::bt_tile(dpl, R, cols, c.row, c.col, wide, tile, tile_hi);
				prof.tiles++; prof.tile_t += now() - tt_;
			}
			if ((pred ? Plat::lane(tile_hi, 0) : Plat::lane(tile, 48)) & 1) { HOT.cural++; continue; }
			// reseeding protocol: 8-bit kernels init(reseed) ... init(reseed+1); 16-bit kernels only init(reseed) afterwards
			// (aligner_sw.cpp:796-933 end-to-end, :962-1110 local)
			const uint32_t reseed = ST.rnd.nextU32() + 1;
			if (!sse16) ST.rnd.init(reseed);
			res.nned = 0;
			const int32_t cscore = c.score;
			bool ret;
			const uint32_t need_c0 = Plat::uni((c.col + 1u > kRfWin) ? (((uint32_t)c.col + 1u - kRfWin + 3u) & ~3u) : 0u);
			if (need_c0 > 0 && rows + 250u > kRfWin - 4u) { ovf(17); ret = false; }   // the walk could leave the window the registers hold (rows + read gaps)
			else {
				if (need_c0 != rf_c0_) { rf_c0_ = need_c0; for (uint32_t k = 0; k < kRfRegs; k++) rfw[k] = Plat::lanes_load(Plat::rf(), ST.max_cols + 8u, k * 64 + (need_c0 >> 2)); }
				const uint64_t tw_ = now();
				ret = walk(c.row, c.col, tile, tile_hi);
#ifdef BT2G_SAMP_STATS
				if (MODE == 0) { static unsigned long sw_[3] = {0, 0, 0}, n_ = 0; sw_[ret ? 0 : 1]++; n_++; if ((n_ & (n_ - 1)) == 0) fprintf(stderr, "WALKSTAT step-loop walks %lu: succeeded %lu failed %lu\n", n_, sw_[0], sw_[1]); }
#endif
				if (PRM.profile) { const uint64_t dt_ = now() - tw_; HOT.t_bt[0] += dt_; if (ret) { HOT.t_bt[1] += dt_; HOT.t_bt[2]++; } }
			}
			ST.rnd.init(sse16 ? reseed : reseed + 1);
			if (MODE == 2) {       // btncanddone_: tried, succeeded or not
				const uint32_t v_ = (uint32_t)c.row | ((uint32_t)c.col << 16);
				if (ndone >= (uint32_t)kMaxCandDone) ovf(33);
				else { gst(donel + ndone, v_); if (ndone < 64u) Plat::set_lane(dn0, ndone, v_); else if (ndone < 128u) Plat::set_lane(dn1, ndone - 64u, v_); ndone++; HOT.n_cdone = ndone; Plat::dom_update(domv, cw1, v_, SQ); }
			}
			(void)cscore;
			if (ret) { found = true; break; }
			HOT.cural++;
			if (MODE == 0) bw_ok = Plat::marks_batchable();
		}
		if (!found) return false;
		if (!fw) invert_edits(res);
		HOT.cural++;
		return true;
	}
	BT2_HD bool next_alignment(bool fw, uint32_t rows, uint32_t cols, const DPRect& rect, uint64_t tidx, int64_t tlen, int mode, bool sse16, BT2_G AlnRes& res) {
		bool r;
		if (mode == 0) r = next_alignment_m<0>(fw, rows, cols, rect.triml, rect.corel, rect.corer, rect.refl, tidx, tlen, sse16, res);
		else if (mode == 1) r = next_alignment_m<1>(fw, rows, cols, rect.triml, rect.corel, rect.corer, rect.refl, tidx, tlen, sse16, res);
		else r = next_alignment_m<2>(fw, rows, cols, rect.triml, rect.corel, rect.corer, rect.refl, tidx, tlen, sse16, res);
		return Plat::uni((int)r) != 0;      // (the result of a real call is lane-varying to the compiler)
	}

	// SwAligner::ungappedAlign, monotone branch (aligner_sw.cpp:286-494); returns 0 / 1
	BT2_HDN int ungapped_align(bool fw_, uint64_t tidx_, int64_t refoff_, int64_t reflen_, BT2_G AlnRes& res_) {
		BT2_G AlnRes& res = *Plat::uni_ptr(&res_);
		const bool fw = Plat::uni((int)fw_) != 0; const uint64_t tidx = Plat::uni(tidx_); const int64_t refoff = Plat::uni(refoff_), reflen = Plat::uni(reflen_);
		const uint32_t len = HOT.len;
		const int64_t rfi = refoff, rff = refoff + (int64_t)len;
		// overhanging ends are only scored (as Ns) with --overhang, and count against the N ceiling (aligner_sw.cpp:306-325)
		if ((rfi < 0 || rff > reflen) && !PRM.overhang) return 0;
		if ((rfi < 0 ? -rfi : 0) + (rff > reflen ? rff - reflen : 0) > (int64_t)RPR.nceil) return 0;
		int64_t score = 0;
		int ns = 0;
		Plat::fetch_ref_codes(IX.ref, tidx, rfi, len);   // codes here, not masks
		uint32_t rowi = 0, rowf = len - 1;
		auto step = [&](uint32_t i) {
			const int rdc = rd_char(HOT, HOT.len, fw, i);
			const int rfc = Plat::rf()[i];
			const int q = rd_qual(HOT, HOT.len, fw, i) - 33;
			if (rdc > 3 || rfc > 3) { ns++; score -= PRM.n_pen; }
			else if (rdc == rfc) score += PRM.match_bonus;
			else score -= mm_penalty(PRM, q < 0 ? 0 : q);
		};
		if (PRM.match_bonus == 0) {
			for (uint32_t i = 0; i < len; i++) {
				step(i);
				if (score < ST.minsc || ns > RPR.nceil) return 0;
			}
		} else {
			// local flavour (aligner_sw.cpp:400-436): best-scoring stretch of the diagonal; more than one -> leave it to the DP
			int64_t score_max = 0;
			uint32_t lastfloor = 0, sols = 0;
			rowi = 0xffffffffu;
			for (uint32_t i = 0; i < len; i++) {
				step(i);
				if (score >= ST.minsc && score >= score_max) {
					score_max = score; rowf = i;
					if (rowi != lastfloor) { rowi = lastfloor; sols++; }
				}
				if (score <= 0) { score = 0; lastfloor = i + 1; }
			}
			if (ns > RPR.nceil || score_max < ST.minsc) return 0;
			if (sols > 1) return -1;
			score = score_max;
		}
		uint32_t nned = 0, refns = 0;
		for (uint32_t i = rowi; i <= rowf; i++) {
			const int rdc = rd_char(HOT, HOT.len, fw, i);
			const int rfc = Plat::rf()[i];
			if (rfc > 3 || rdc != rfc) {
				if (nned >= (uint32_t)kMaxEdits) { ovf(22); return 0; }
				Edit& e = res.ned[nned++];
				e.pos = (uint16_t)i; e.chr = code2chr(rfc); e.qchr = code2chr(rdc); e.type = EDIT_MM;
				if (rfc > 3) refns++;
			}
		}
		res.nned = (uint16_t)nned;
		res.score = (int32_t)score; res.ns = (int16_t)ns; res.gaps = 0; res.edits = (int16_t)nned;
		res.bases_aligned = (int16_t)((int)len - (int)nned);
		res.refns = (uint16_t)refns;
		const uint32_t trim_end = (len - 1) - rowf;
		set_shape(res, (int32_t)tidx, refoff + (int64_t)rowi, reflen, fw, len, fw ? rowi : trim_end, fw ? trim_end : rowi);
		if (!fw) invert_edits(res);
		return 1;
	}

	// =================================================================================
	// F. SwDriver::extendSeeds (aligner_sw_driver.cpp:921-1494)
	// =================================================================================
	BT2_HDI int extend_seeds(int seedmms_, int seedlen, int seedival) {
		(void)seedlen; (void)seedival;
		const int seedmms = Plat::uni(seedmms_);
		// (the phase timers of this function ask one scalar register whether anybody is listening, not the parameter block in LDS at every site)
		const bool prof_ = Plat::uni((int)PRM.profile) != 0;
		auto tnow = [prof_]() -> uint64_t { return prof_ ? Plat::clock() : 0ull; };
		// (What this function keeps across its calls are wave-uniform scalars.  A value read from LDS arrives in a vector register and stays in one
		// -- 64 lanes for one number, spilled to scratch around every call: a memory round trip per use -- unless it is said to be uniform.)
		const uint32_t rdlen = Plat::uni(HOT.len);
		const int64_t perfect = (int64_t)rdlen * Plat::uni(PRM.match_bonus);       // Scoring::perfectScore: 0 end to end
		const uint32_t nsm = 5;
		const uint32_t nonz = Plat::uni(HOT.nonz_tot);
		const uint64_t ee_hits = Plat::uni((uint64_t)((HOT.exact[0].bot - HOT.exact[0].top) + (HOT.exact[1].bot - HOT.exact[1].top) + HOT.mm1_elt));
		bool ee_mode = ee_hits > 0;
		bool first_ee = true, first_extend = true;
		HOT.n_ee_fail = HOT.n_ug_fail = HOT.n_dp_fail = 0;
		uint64_t nelt = 0, nelt_left = 0;
		const uint32_t rows = rdlen;
		const uint32_t max_iters = Plat::uni((uint32_t)PRM.max_iters);
		BT2_G AlnRes& res = WK.res;
		// A sampled row (most entries of the extension list on repeats) is ONE row of a range: what the loop needs of it -- strand, seed offset and
		// length, seed number -- are its range's fields, packed into one lane register (lane j = range samp_sai + j) when the sampler had at most 64
		// ranges; its row and "taken" flag come in one 16-byte load.  (Expanding every such row into a SatPos with a Random1toN of one element in the
		// arena, as rounds 1-5 did, cost six dependent memory round trips per row: 83 us per read of the headline workload.)
		typename Plat::LaneReg rgpk;
		Plat::lanes_zero(rgpk);
		bool lane_rows = false;
		while (true) {
			if (ee_mode) {
				if (first_ee) {
					first_ee = false;
					ee_sa_tups(nelt, max_iters);
					ee_mode = true;
				} else ee_mode = false;
			}
			if (!ee_mode) {
				if (nonz == 0) return EXT_EXHAUSTED;
				if (Plat::uni(ST.minsc) == perfect) return EXT_PERFECT_SCORE;
				if (first_extend) {
					nelt = 0;
					{ const uint64_t t0_ = tnow(); prioritize(seedmms, max_iters, nelt); HOT.t_phase[3] += tnow() - t0_; }
					nelt = Plat::uni(nelt);
					nelt_left = nelt;
					first_extend = false;
					lane_rows = Plat::uni((int)(!PRM.det_seeds && HOT.samp_lanes != 0u && HOT.n_satpos > HOT.n_satpos_full)) != 0;
					if (lane_rows) rgpk = Plat::range_fields(&WK.satpos2[HOT.samp_sai], HOT.n_masses);
				}
				if (nelt_left == 0) break;
			}
			const uint32_t maxi = Plat::uni(HOT.n_satpos), n_full = Plat::uni(HOT.n_satpos_full), samp_sai = Plat::uni(HOT.samp_sai);
			for (uint32_t i = 0; i < maxi; i++) {
#ifdef BT2G_LOOP_PROF
				uint64_t tl_ = Plat::clock();
#define LOOP_T(slot) do { const uint64_t t__ = Plat::clock(); HOT.t_phase[slot] += t__ - tl_; tl_ = t__; } while (0)
#else
#define LOOP_T(slot) do {} while (0)
#endif
				// the entry: a whole range / an end-to-end hit with its Random1toN in the arena, or one sampled row
				const bool srow = lane_rows && i >= n_full;
				BT2_G SatPos* spp = nullptr;
				uint64_t sp_topf; uint32_t sp_size, sp_rdoff, sp_seedlen, sp_offidx, sp_orig_sz; int32_t sp_ee; bool sp_fw;
				bool srow_done = false;
				if (srow) {
					const SampRow sr = gld(&WK.srows[i - n_full]);
					if (Plat::uni(sr.done) != 0u) continue;
					const uint32_t pk = Plat::lane(rgpk, Plat::uni(sr.src) - samp_sai);
					sp_topf = Plat::uni(sr.topf); sp_size = 1u; sp_orig_sz = 2u; sp_ee = -1;
					sp_rdoff = pk & 0xfffu; sp_seedlen = (pk >> 12) & 0x3fu; sp_fw = ((pk >> 18) & 1u) != 0u; sp_offidx = pk >> 20;
				} else {
					if (satpos_taken(i)) continue;
					spp = &satpos_view(i);
					spp = Plat::uni_ptr(spp);
					sp_topf = Plat::uni((uint64_t)spp->topf); sp_size = Plat::uni((uint32_t)spp->size); sp_orig_sz = Plat::uni((uint32_t)spp->orig_sz); sp_ee = Plat::uni((int)spp->ee);
					sp_rdoff = Plat::uni((uint32_t)spp->rdoff); sp_seedlen = Plat::uni((uint32_t)spp->seedlen); sp_fw = Plat::uni((int)(spp->fw != 0)) != 0; sp_offidx = Plat::uni((uint32_t)spp->offidx);
				}
				LOOP_T(0);
				EEHit eh_v; eh_v.score = 0;
				if (ee_mode) eh_v = ee_hit(sp_ee);
				const EEHit* eh = &eh_v;
				if (ee_mode && eh->score < Plat::uni(ST.minsc)) return EXT_PERFECT_SCORE;
				const bool is_small = Plat::uni((int)PRM.det_seeds) ? true : sp_size < nsm;
				const bool fw = sp_fw;
				uint32_t rdoff = sp_rdoff;
				const uint32_t seedhitlen = sp_seedlen;
				if (!fw) rdoff = rdlen - rdoff - seedhitlen;
				bool first = true;
				while (!(srow ? srow_done : r1n_done(spp->rnd)) && (first || is_small || ee_mode)) {
					const int64_t minsc_now = Plat::uni(ST.minsc);
					if (minsc_now == perfect) {
						if (!ee_mode || eh->score < perfect) return EXT_PERFECT_SCORE;
					} else if (ee_mode && eh->score < minsc_now) {
						break;
					}
					if (Plat::uni(HOT.n_ex_dps) >= Plat::uni((uint32_t)PRM.max_dp)) return EXT_HARD_LIMIT;
					if (Plat::uni(HOT.n_ex_ugs) >= Plat::uni((uint32_t)PRM.max_ug)) return EXT_HARD_LIMIT;
					if (Plat::uni(HOT.n_ex_iters) >= max_iters) return EXT_HARD_LIMIT;
					HOT.n_ex_iters++;
					first = false;
#ifdef BT2G_ITER_PROF
					// diagnostic build: where the extension loop's own time goes -- draws that end as "seen this diagonal" / "not in a sequence" (slot 0, count x 100 in
					// slot 16) against draws that go on to a DP problem (slot 15, the named phases included; all draws x 100 in slot 1)
					struct IterT { uint64_t t0; bool skip; BT2_HD ~IterT() { const uint64_t d = Plat::clock() - t0; if (skip) { HOT.t_phase[0] += d; HOT.t_phase[16] += 100; } else HOT.t_phase[15] += d; HOT.t_phase[1] += 100; } } iter_t_{Plat::clock(), false};
#define ITER_SKIP() (iter_t_.skip = true)
#else
#define ITER_SKIP() do {} while (0)
#endif
					uint32_t elt = 0;      // (a sampled row: Random1toN of one element -- no draw, random_util.h:90-94)
					if (srow) srow_done = true; else elt = Plat::uni(r1n_next(spp->rnd));
					// GroupWalk2S::advanceElement == Ebwt::getOffset(topf + elt)
					uint32_t steps = 0;
					const uint64_t tr_ = tnow();
					TOff joff;
					uint64_t jc = kJoffNone;
					// a one-row hit of the pre-computed seed round was resolved by the batch kernel that extended it
					if (!ee_mode && ST.ext_pre && seedmms == 0 && sp_orig_sz == 1 && ST.pre_joff_cur)
						jc = Plat::uni((uint64_t)Plat::uni_ptr(ST.pre_joff_cur)[((uint64_t)Plat::uni(ST.ridx) * 2 + (fw ? 0 : 1)) * Plat::uni(PRE->max_seeds) + sp_offidx]);
					if (jc == kJoffNone && !ee_mode && i >= n_full) {
						// a sampled row: the walks to the SA sample of this row and the next 63 run side by side, one per lane
						// (the extension loop takes the rows in list order, so the look-ahead is rarely wasted)
						if (i >= Plat::uni(HOT.n_resolved)) {
							const uint32_t k0 = i - n_full, cnt = maxi - i < 64u ? maxi - i : 64u;
							Plat::resolve_rows(IX.fw, &WK.srows[k0], cnt, &WK.srow_joff[k0]);
							HOT.n_resolved = i + cnt;
						}
						jc = Plat::uni((uint64_t)WK.srow_joff[i - n_full]);
						if (jc != kJoffNone) HOT.n_sides += (uint32_t)(jc >> 48);
					}
					if (jc != kJoffNone) { joff = (TOff)(jc & 0xffffffffffffull); steps = (uint32_t)(jc >> 48); }
					else { joff = Plat::get_offset(IX.fw, (TOff)(sp_topf + elt), steps); HOT.n_sides += steps; }
					HOT.t_phase[4] += tnow() - tr_;
					HOT.n_bwops_ext += steps; HOT.n_resolve_steps += steps;
					if (!ee_mode) nelt_left--;
					TOff tidx = 0, toff = 0, tlen = 0;
					bool straddled = false;
#ifdef BT2G_LOOP_PROF
					tl_ = Plat::clock();
#endif
					Plat::joined_to_text(IX, (TOff)seedhitlen, joff, tidx, toff, tlen, ee_mode, straddled);
					tidx = (TOff)Plat::uni((uint64_t)tidx); toff = (TOff)Plat::uni((uint64_t)toff); tlen = (TOff)Plat::uni((uint64_t)tlen);      // (shuffle results are lane-varying to the compiler)
					LOOP_T(1);
					if (tidx == kOffMask) { ITER_SKIP(); continue; }
					const int64_t refoff = (int64_t)toff - (int64_t)rdoff;
					{ const bool dp_ = diag_present((int32_t)tidx, refoff, fw); LOOP_T(15); if (dp_) { HOT.n_redundants++; ITER_SKIP(); continue; } }
					int read_gaps = 0, ref_gaps = 0;
					bool ungapped = false;
					if (!ee_mode) {
						read_gaps = Plat::uni(max_read_gaps(PRM, minsc_now, rdlen));
						ref_gaps = Plat::uni(max_ref_gaps(PRM, minsc_now, rdlen));
						ungapped = (read_gaps == 0 && ref_gaps == 0);
					}
					int state = 0;   // 0 none, 1 ee, 2 ungapped
					int mode = 0;        // cell format of this DP (see backtrace)
					bool sse16 = false;  // the reference's 16-bit kernel produced this matrix (RNG protocol of nextAlignment)
					uint32_t lastsolcol = 0;
					bool found = false;
					DPRect rect;
					rect.refl = rect.refr = rect.refl_pretrim = rect.refr_pretrim = 0;
					rect.triml = rect.trimr = rect.corel = rect.corer = rect.maxgap = 0;
					uint32_t cols = 0;
					if (ee_mode) {
						res.nned = 0;
						const int mms = eh->has_edit ? 1 : 0;
						int hns = 0, hrefns = 0;
						if (mms) {
							hns = (eh->echr == 4 || eh->eqchr == 4) ? 1 : 0;
							hrefns = (eh->echr == 4) ? 1 : 0;
						}
						res.score = eh->score; res.bases_aligned = (int16_t)((int)rdlen - mms); res.edits = (int16_t)mms;
						res.ns = (int16_t)hns; res.gaps = 0;
						if (mms) {
							Edit& e = res.ned[0];
							e.pos = eh->epos; e.chr = code2chr(eh->echr); e.qchr = code2chr(eh->eqchr); e.type = EDIT_MM; e.pad = 0;
							res.nned = 1;
						}
						// setShape with no trimming leaves the (already 5'-relative) edit untouched
						set_shape(res, (int32_t)tidx, refoff, (int64_t)tlen, fw, rdlen, 0, 0);
						res.refns = (uint16_t)hrefns;
						state = 1; found = true;
						diag_add((int32_t)tidx, refoff, fw, 1);
					} else if (ungapped && Plat::uni((int)PRM.do_ungapped)) {
						const uint64_t tu_ = tnow();
						const int al = Plat::uni(ungapped_align(fw, tidx, refoff, (int64_t)tlen, res));
						HOT.t_phase[10] += tnow() - tu_;
						diag_add((int32_t)tidx, refoff, fw, 1);
						HOT.n_ex_ugs++;
						if (al == 0) {
							HOT.n_ug_fail++;
							if (HOT.n_ug_fail >= (uint32_t)PRM.max_dp_streak) return EXT_SOFT_LIMIT;
							continue;
						}
						if (al == -1) {           // several equally good stretches on this diagonal: count a failure, let the DP decide (:1250-1256)
							HOT.n_ug_fail++;
							if (HOT.n_ug_fail >= (uint32_t)PRM.max_dp_streak) return EXT_SOFT_LIMIT;
						} else {
							HOT.n_ug_fail = 0;
							found = true; state = 2;
						}
					}
					if (state == 0) {
						// DynProgFramer::frameSeedExtensionRect (dp_framer.cpp:81-129), trimToRef
						uint32_t maxgap = (uint32_t)imax(read_gaps, ref_gaps);
						{ const uint32_t mh_ = Plat::uni((uint32_t)PRM.maxhalf); if (maxgap > mh_) maxgap = mh_; }
						const int64_t refl = refoff - 2 * (int64_t)maxgap;
						const int64_t refr = refoff + ((int64_t)rows - 1) + 2 * (int64_t)maxgap;
						uint64_t triml = 0, trimr = 0;
						// trimToRef_ = !gReportOverhangs; otherwise up to nceil columns of N past either end stay in the window
						int64_t maxns = 0;
						if (Plat::uni((int)PRM.overhang)) { maxns = Plat::uni((int)RPR.nceil); if (maxns == (int64_t)rows) maxns--; }
						if (refr >= (int64_t)tlen + maxns) trimr = (uint64_t)(refr - ((int64_t)tlen + maxns - 1));
						if (refl < -maxns) triml = (uint64_t)(-refl) - (uint64_t)maxns;
						rect.refl_pretrim = refl; rect.refr_pretrim = refr;
						rect.refl = refl + (int64_t)triml; rect.refr = refr - (int64_t)trimr;
						rect.triml = (uint32_t)triml; rect.trimr = (uint32_t)trimr; rect.maxgap = maxgap;
						rect.corel = maxgap; rect.corer = rect.corel + 2 * maxgap;
						found = !(rect.refr < rect.refl);
						diag_add((int32_t)tidx, refoff, fw, 1);
						if (!found) continue;
						cols = (uint32_t)(rect.refr - rect.refl + 1);
						if (cols + 1 > Plat::uni(ST.max_cols) || rows > (uint32_t)kMaxLen) { ovf(23); return EXT_HARD_LIMIT; }
						diag_add((int32_t)tidx, rect.refl_pretrim + (int64_t)rect.corel, fw, (int64_t)(rect.corer - rect.corel + 1));
						// SwAligner::align (aligner_sw.cpp:500-729).  End to end: 8-bit kernel while ST.minsc >= -254, else 16-bit (:517).
						// Local: the 8-bit kernel unless it saturates, then the 16-bit one (:568-600); the fill below is exact and
						// reports whether the 8-bit kernel would have saturated, which only matters for the RNG protocol.
						const uint64_t td_ = tnow();
						fetch_ref_window(tidx, rect.refl, cols + 1);
						int64_t best;
						if (Plat::uni(PRM.match_bonus) > 0) {
							mode = 2;
							uint32_t sat8 = 0;
							best = Plat::uni(Plat::dp_fill_local(PRM, WK, fw, rows, cols, ST.dp.mat, minsc_now, lastsolcol, sat8));
							lastsolcol = Plat::uni(lastsolcol);
							sse16 = Plat::uni(sat8) != 0;
						} else {
							// below -254 the reference's 16-bit kernel: the device fills it on the band like the 8-bit one wherever the band fits (matrix format 0,
							// the 16-bit kernel's RNG protocol), else -- and on the CPU twin -- in the anti-diagonal cell format (1)
							sse16 = minsc_now < -254;
							mode = (sse16 && !Plat::ee_wide_band(rows, cols, minsc_now)) ? 1 : 0;
							best = Plat::uni(Plat::dp_fill_ee(PRM, WK, fw, rows, cols, ST.dp, sse16, minsc_now));
							if (best == INT64_MIN) { ovf(31); return EXT_HARD_LIMIT; }
						}
						HOT.t_phase[5] += tnow() - td_;
						HOT.n_ex_dps++;
						found = best >= minsc_now;
#ifdef BT2G_COUNT_FAILED_FILLS
						if (!found) HOT.t_phase[15] += 100;      // diagnostic build: DP windows whose best score stays below the minimum (per read, in the "gather_lastrow" slot)
#endif
						if (found) { const uint64_t tg_ = tnow(); gather_cells(fw, rows, cols, minsc_now, mode, lastsolcol); found = Plat::uni(HOT.n_cands) > 0; HOT.t_phase[8] += tnow() - tg_; }
						if (!found) {
							HOT.n_dp_fail++;
							if (Plat::uni(HOT.n_dp_fail) >= Plat::uni((uint32_t)PRM.max_dp_streak)) return EXT_SOFT_LIMIT;
							continue;
						}
						if (HOT.n_dp_fail > HOT.n_dp_fail_streak) HOT.n_dp_fail_streak = HOT.n_dp_fail;
						HOT.n_dp_fail = 0;
					}
					bool first_inner = true;
					while (true) {
						if (state == 1 || state == 2) {
							if (!first_inner) break;
						} else {
							if (Plat::uni(HOT.cural) == Plat::uni(HOT.n_cands)) break;
							const uint64_t tb_ = tnow();
							const bool na_ = next_alignment(fw, rows, cols, rect, tidx, (int64_t)tlen, mode, sse16, res);
							HOT.t_phase[6] += tnow() - tb_;
							if (!na_) break;
						}
						first_inner = false;
						const uint64_t tp_ = tnow();
						struct PostTimer { uint64_t t0; uint64_t& acc; bool on; BT2_HD ~PostTimer() { if (on) acc += Plat::clock() - t0; } } post_timer_{tp_, HOT.t_phase[9], prof_};
						// --overhang: soft-clip what hangs off either end (aligner_sw_driver.cpp:1396-1403)
						if (PRM.overhang && (res.refoff < 0 || res.refoff + (int64_t)res.rfextent > (int64_t)tlen)) {
							clip_outside(res, 0, (int64_t)tlen);
							if (res.rfextent == 0) continue;
						}
						// fell entirely outside the reference?
						{
							const int64_t a0 = res.refoff, a1 = res.refoff + res.rfextent;
							const int64_t b0 = 0, b1 = (int64_t)tlen;
							const bool ov = (b0 <= a0 && b1 > a0) || (b0 <= a1 && b1 > a1) || (a0 <= b0 && a1 > b0) || (a0 <= b1 && a1 > b1);
							if (!ov) continue;
						}
						if (Plat::uni((int)red_overlap(res)) != 0) continue;
						red_add(res);
						{ const uint64_t t1_ = tnow(); const bool sr_ = sink_report(res); HOT.t_phase[19] += tnow() - t1_; if (sr_) return EXT_POLICY_FULFILLED; }
						if (PRM.tighten > 0 && PRM.mhits > 0 && HOT.best2_unp1 != INT64_MIN) {
							if (PRM.tighten == 1) {
								if (HOT.best_unp1 >= ST.minsc) {
									ST.minsc = HOT.best_unp1;
									if (ST.minsc < perfect && HOT.best_unp1 == HOT.best2_unp1) ST.minsc++;
								}
							} else if (PRM.tighten == 2) {
								if (HOT.best2_unp1 >= ST.minsc) { ST.minsc = HOT.best2_unp1; if (ST.minsc < perfect) ST.minsc++; }
							} else {
								const int64_t diff = HOT.best_unp1 - HOT.best2_unp1;
								const int64_t bot = HOT.best2_unp1 + ((diff * 3) / 4);
								if (bot >= ST.minsc) { ST.minsc = bot; if (ST.minsc < perfect) ST.minsc++; }
							}
						}
					}
				}
#ifdef BT2G_LOOP_PROF
				tl_ = Plat::clock();
#endif
				if (srow) gst(&WK.srows[i - HOT.n_satpos_full].done, srow_done ? 1u : 0u); else satpos_commit(i, *spp);
				LOOP_T(16);
			}
			if (PRM.det_seeds) break;      // useCurrIdx: always one pass (aligner_sw_driver.cpp:1490)
		}
		return EXT_EXHAUSTED;
	}

	// =================================================================================
	// G. the per-read worker (multiseedSearchWorker, bt2_search.cpp:3094-4254, unpaired path)
	// =================================================================================
	BT2_HD void handle_ret(int ret, bool& done) {
		if (ret == EXT_POLICY_FULFILLED) { if (HOT.done_unpair1) done = true; }
		else if (ret == EXT_PERFECT_SCORE) done = true;
		else if (ret == EXT_HARD_LIMIT) done = true;
	}

	BT2_HD void run(BT2_G ReadResult& out) {
		const uint32_t len = HOT.len;
		HOT.err = 0;
		HOT.n_alns = 0; HOT.best_unp1 = HOT.best2_unp1 = INT64_MIN; HOT.done_unpair1 = 0; HOT.exit_m = HOT.exit_k = 0;
		HOT.n_diags = 0; HOT.n_ex_fw = HOT.n_ex_rc = 0;
		HOT.n_ex_iters = HOT.n_ex_dps = HOT.n_ex_ugs = HOT.n_dp_fail = HOT.n_ug_fail = HOT.n_ee_fail = HOT.n_dp_fail_streak = 0;
		HOT.n_redundants = HOT.n_bwops_seed = HOT.n_bwops_ext = HOT.n_bt_attempts = 0; HOT.n_sides = 0; HOT.n_ext_left = HOT.n_ext_right = HOT.n_resolve_steps = 0;
		HOT.n_dp_cells_score = HOT.n_dp_cells_full = HOT.n_dp_pass = 0;
		HOT.frag_tidx = ~0ull; HOT.frag_len = 0;
		for (int i_ = 0; i_ < 22; i_++) HOT.t_phase[i_] = 0;
		for (int i_ = 0; i_ < 5; i_++) HOT.t_bt[i_] = 0;
		const uint64_t t_run0_ = now();
		HOT.n_mm1 = 0; HOT.mm1_elt = 0; HOT.nonz_tot = 0; HOT.n_rank = 0; HOT.num_offs = 0; HOT.num_elts = 0;
		HOT.exact[0].top = HOT.exact[0].bot = HOT.exact[1].top = HOT.exact[1].bot = 0;
		ST.minsc = RPR.minsc;
		const bool filt = (RPR.filt & 15u) == 15u;
		bool done = !filt;
		const int64_t perfect = (int64_t)len * PRM.match_bonus;
		if (!done) {
			ST.rnd.init(RPR.seed);
			const uint32_t interval = (uint32_t)RPR.interval;
			uint32_t nrounds = (uint32_t)PRM.n_seed_rounds;
			if (nrounds > interval) nrounds = interval;
			uint32_t mine[2] = {0, 0};
			uint64_t nelt = 0;
			// The three stages of the worker -- exact end-to-end hits, 1-mismatch end-to-end hits, the seeding rounds -- all end in
			// extendSeeds.  They run as the iterations of ONE loop so that extend_seeds has a single call site and is inlined into the
			// kernel: as a real function it saved and restored every callee-saved vector register in its prologue and epilogue and spilled
			// what it kept across its own calls (a 900-byte frame per lane, ~60 KB of scratch traffic per call and direction).
			const uint32_t n_stage = 2u + (uint32_t)PRM.n_seed_rounds;
			for (uint32_t stage = 0; stage < n_stage; stage++) {
				int ext_mms = -1;
				if (stage == 0) {
					if (!PRM.do_exact_upfront) continue;
					{ const uint64_t t0_ = now(); nelt = (PRE && PRE->sweep) ? exact_sweep_pre(mine) : exact_sweep(2, mine); HOT.t_phase[0] += now() - t0_; }
					if (nelt == 0) { HOT.exact[0].top = HOT.exact[0].bot = HOT.exact[1].top = HOT.exact[1].bot = 0; continue; }
				} else if (stage == 1) {
					if (!PRM.do_1mm_upfront) continue;
					nelt = 0;
					if (!done) {
						const bool yfw = mine[0] <= 1 && !ST.m_nofw;
						const bool yrc = mine[1] <= 1 && !ST.m_norc;
						if (yfw || yrc) {
							const uint64_t t0_ = now();
							if (!(PRE && PRE->mm1 && one_mm_pre(!yfw, !yrc))) one_mm_search(!yfw, !yrc);
							nelt = HOT.mm1_elt; HOT.t_phase[1] += now() - t0_;
						}
					}
					if (nelt == 0) { HOT.n_mm1 = 0; HOT.mm1_elt = 0; continue; }
				} else {
					const uint32_t roundi = stage - 2u;
					HOT.nonz_tot = 0; HOT.n_rank = 0; HOT.num_elts = 0; HOT.num_offs = 0;
					if (done || HOT.done_unpair1) { done = true; continue; }
					if (roundi >= nrounds) continue;
					if (interval <= roundi) continue;
					const uint32_t offset = (interval * roundi) / nrounds;
					if (offset > 0 && (uint32_t)RPR.seedlen + offset > len) continue;
					const uint64_t ts_ = now();
					ST.ext_pre = false;
					cache_reset();          // ca.nextRead() (bt2_search.cpp:3882)
					uint32_t ninst;
					if (PRM.seed_mms > 0) ninst = Plat::uni(seed_round_mm1(offset, interval, (uint32_t)RPR.seedlen));
					else if (offset == 0 && PRE && PRE->seeds && 1 + (len > (uint32_t)RPR.seedlen ? (len - (uint32_t)RPR.seedlen) / interval : 0u) <= PRE->max_seeds) {
						ninst = seed_round_pre(PRE->seeds, 0, interval, (uint32_t)RPR.seedlen);
						ST.ext_pre = PRE->ext != nullptr; ST.pre_ext_cur = PRE->ext; ST.pre_joff_cur = PRE->joff;
					} else if (roundi > 0 && roundi < kMaxPreRounds && PRE && PRE->seeds_r[roundi] && 1 + (len > offset + (uint32_t)RPR.seedlen ? (len - offset - (uint32_t)RPR.seedlen) / interval : 0u) <= PRE->max_seeds &&
					           PRE->seeds_r[roundi][(uint64_t)ST.ridx * 2 * PRE->max_seeds].topf != ~0ull) {
						// a re-seeding round the batch kernels searched (they saw the same "previous round was repetitive" condition:
						// a read only gets here when it held); a read with more seed positions than the tables hold searches them itself
						ninst = seed_round_pre(PRE->seeds_r[roundi], offset, interval, (uint32_t)RPR.seedlen);
						ST.ext_pre = PRE->ext_r[roundi] != nullptr; ST.pre_ext_cur = PRE->ext_r[roundi]; ST.pre_joff_cur = PRE->joff_r[roundi];
					} else ninst = Plat::uni(seed_round(offset, interval, (uint32_t)RPR.seedlen));
					HOT.t_phase[2] += now() - ts_;
					if (ninst == 0) { done = true; HOT.nonz_tot = 0; continue; }
					if (HOT.nonz_tot == 0) { done = true; continue; }
					{ const uint64_t t0_ = now(); rank_seed_hits(); HOT.t_phase[3] += now() - t0_; }
					ext_mms = PRM.seed_mms;
				}
#ifdef BT2G_ITER_PROF
				const uint64_t te_ = Plat::clock();
				const int ret = Plat::uni(extend_seeds(ext_mms, RPR.seedlen, (int)interval));
				HOT.t_phase[20] += Plat::clock() - te_;
#else
				const int ret = Plat::uni(extend_seeds(ext_mms, RPR.seedlen, (int)interval));
#endif
				handle_ret(ret, done);
				if (stage == 0) {
					HOT.exact[0].top = HOT.exact[0].bot = HOT.exact[1].top = HOT.exact[1].bot = 0;
					if (!done && ST.minsc == perfect) done = true;
				} else if (stage == 1) {
					HOT.n_mm1 = 0; HOT.mm1_elt = 0;
					if (!done && ST.minsc == perfect) done = true;
				} else {
					if (!done && HOT.nonz_tot > 0 && (HOT.num_elts / HOT.nonz_tot) < (uint64_t)PRM.seed_boost_thresh) done = true;
				}
			}
		}
#ifdef BT2G_LOOP_PROF
		{ const uint64_t tf_ = Plat::clock(); finish(out); HOT.t_phase[20] += Plat::clock() - tf_; }
#else
		finish(out);
#endif
		HOT.t_phase[11] = ST.pf_steps; HOT.t_phase[12] = ST.pf_tiles; HOT.t_phase[14] = ST.pf_tile_t;
		HOT.t_phase[7] = now() - t_run0_;
#ifdef BT2G_DIAG_TICKS
		out.n_ext_left = (uint32_t)HOT.t_phase[7]; out.n_ext_right = (uint32_t)HOT.t_phase[5]; out.n_resolve_steps = (uint32_t)HOT.t_phase[6]; out.n_sides = (uint32_t)HOT.t_phase[3];      // diagnostic build: whole read / dp fill / backtrace / rank+prioritise ticks per read
#endif
#ifdef BT2G_DEBUG_SATPOS
		{
			uint32_t* dbg = reinterpret_cast<uint32_t*>(out.alns[0].ned);
			uint32_t k = 0;
			dbg[k++] = HOT.n_satpos2;
			for (uint32_t i = 0; i < HOT.n_satpos2 && k + 6 < 290; i++) {
				const BT2_G SatPos& s = WK.satpos2[i];
				dbg[k++] = (uint32_t)s.topf; dbg[k++] = (uint32_t)s.topb; dbg[k++] = s.size; dbg[k++] = s.nlex; dbg[k++] = s.nrex; dbg[k++] = s.offidx * 2 + s.fw;
			}
		}
#endif
	}

	// AlnSinkWrap::finishRead for an unpaired read (aln_sink.cpp:643-1384): ReportingState::finish,
	// getReport, selectByScore (RNG!), and what the SAM line needs.
	BT2_HDN void finish(BT2_G ReadResult& out_) {
		BT2_G ReadResult& out = *Plat::uni_ptr(&out_);
		// -a: the reference reports every alignment found; this build's result record holds khits (BT2G_MAX_KHITS) of them
		if (PRM.all_hits && HOT.n_alns > (uint32_t)PRM.khits) ovf(32);
		out.status = (uint8_t)HOT.err;
		out.filt = (uint8_t)RPR.filt;
		out.exhausted = 0;
		out.nalns = HOT.n_alns;
		out.n_ex_iters = HOT.n_ex_iters; out.n_ex_dps = HOT.n_ex_dps; out.n_ex_ugs = HOT.n_ex_ugs;
		out.n_dp_fail_streak_max = HOT.n_dp_fail_streak; out.n_bwops_seed = HOT.n_bwops_seed; out.n_bwops_ext = HOT.n_bwops_ext;
		out.n_redundants = HOT.n_redundants; out.n_bt_attempts = HOT.n_bt_attempts;
		out.n_ext_left = HOT.n_ext_left; out.n_ext_right = HOT.n_ext_right; out.n_resolve_steps = HOT.n_resolve_steps; out.n_sides = HOT.n_sides;
		uint32_t nunpair1 = 0;
		bool maxed = false;
		if (HOT.n_alns > 0) {
			if (HOT.exit_k) nunpair1 = (uint32_t)PRM.khits;
			else if (HOT.exit_m) { maxed = true; nunpair1 = 1; }
			else nunpair1 = HOT.n_alns < (uint32_t)PRM.khits ? HOT.n_alns : (uint32_t)PRM.khits;
		}
		out.aligned = nunpair1 > 0 ? 1 : 0;
		out.maxed = maxed ? 1 : 0;
		out.has_secbest = 0; out.secbest = 0; out.best = 0; out.nreport = 0;
		out.pair_type = 0; out.pair_flags = 0; out.pair_best = 0; out.pair_secbest = 0; out.n_mate_dps = 0; out.pad2 = (HOT.err >> 8) | kAlnSlotTag;
		if (nunpair1 == 0) return;
		// selectByScore: sort (score, index) ascending, reverse, shuffle equal-score streaks
		const uint32_t sz = HOT.n_alns < (uint32_t)kMaxAlns ? HOT.n_alns : (uint32_t)kMaxAlns;
		uint32_t num = nunpair1 < sz ? nunpair1 : sz;
		BT2_G uint32_t* idx = WK.lists;      // scratch (Random1toN lists are dead by now)
		if (sz > 64u) Plat::order_by_score(WK.alns, sz, idx, idx + sz);      // (-k / -a with many alignments: every lane ranks its own)
		else {
		for (uint32_t i = 0; i < sz; i++) idx[i] = i;
		for (uint32_t i = 1; i < sz; i++) {          // descending by (score, index)
			const uint32_t v = idx[i];
			uint32_t j = i;
			while (j > 0 && (WK.alns[idx[j - 1]].score < WK.alns[v].score ||
			                 (WK.alns[idx[j - 1]].score == WK.alns[v].score && idx[j - 1] < v))) { idx[j] = idx[j - 1]; j--; }
			idx[j] = v;
		}
		}
		auto shuffle = [&](uint32_t begin, uint32_t n) {
			if (n < 2) return;
			uint32_t left = n;
			for (uint32_t i = begin; i < begin + n - 1; i++) {
				const uint64_t rndi = ST.rnd.nextU64() % left;
				if (rndi > 0) { const uint32_t t = idx[i]; idx[i] = idx[i + rndi]; idx[i + rndi] = t; }
				left--;
			}
		};
		uint32_t streak = 0;
		for (uint32_t i = 1; i < sz; i++) {
			if (WK.alns[idx[i]].score == WK.alns[idx[i - 1]].score) { if (streak == 0) streak = 1; streak++; }
			else { if (streak > 1) shuffle(i - streak, streak); streak = 0; }
		}
		if (streak > 1) shuffle(sz - streak, streak);
		out.best = WK.alns[idx[0]].score;
		if (sz > 1) { out.has_secbest = 1; out.secbest = WK.alns[idx[1]].score; }
		out.nreport = num;
		for (uint32_t i = 0; i < num; i++) Plat::copy_aln(out.alns[i], WK.alns[idx[i]]);
	}

#ifndef BT2G_NO_PAIRS
#include "bt2g_align_pe.inc"
#else
	BT2_HD BtCand* cand_list() { return ST.cands_cur; }      // (the candidate list in use: always Work::cands in a class without pairs)
#endif
};

} // namespace bt2g
#endif
