/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" wrapper (our code) around the *reference's own* classes
 * (Ebwt, SideLocus, SeedAligner, SwAligner, BitPairReference, RandomSource),
 * compiled against the headers and objects where they lie in /root/reference by
 * oracle/Makefile into oracle/_ref/libbt2ref_{s,l}.so.  It lets the tests pin
 * oracle/bt2_oracle.c (and, transitively, the HIP kernels) function-by-function
 * against the real reference, and lets tests/golden/make_golden.py dump golden
 * vectors.  It contains no reference source; it only calls it.
 *
 * `#define private public` is used to reach SwAligner's filled SSE matrices and
 * SeedAligner's counters -- acceptable for a test shim, never shipped.
 */
// Pull in every standard header first so that the access hack below only
// touches the reference's own classes.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <inttypes.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/mman.h>
#include <sys/shm.h>
#include <zlib.h>
#include <iostream>
#include <iomanip>
#include <sstream>
#include <fstream>
#include <limits>
#include <string>
#include <utility>
#include <stdexcept>
#include <vector>
#include <cassert>
#include <thread>
#include <algorithm>
#include <memory>
#include <condition_variable>
#include <queue>
#include <mutex>
#include <future>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <type_traits>
#include <map>
#include <array>
#include <cmath>
#define private public
#define protected public
#include "bt2_idx.h"
#include "aligner_seed.h"
#include "aligner_sw.h"
#include "aligner_cache.h"
#include "reference.h"
#include "random_source.h"
#include "scoring.h"
#include "simple_func.h"
#include "read.h"
#include "dp_framer.h"
#include "aligner_seed_policy.h"
#undef private
#undef protected

#include <string>
#include <vector>
#include <cstring>
#include <cmath>

// A few globals the reference's translation units expect the main program to define.
// (bt2_search.cpp defines them in the real binary; we are not linking that file.)
int gVerbose = 0;
int gDefaultSeedLen = 22;      // DEFAULT_SEEDLEN
bool gReportOverhangs = false; // bt2_search.cpp default

struct RefCtx {
	Ebwt *fw;
	Ebwt *bw;
	BitPairReference *ref;
	Scoring *sc;
	SimpleFunc scoreMin, nCeil;
};

static Scoring *make_default_scoring(RefCtx *c) {
	// defaults of resetOptions (bt2_search.cpp:303-502) for end-to-end mode
	c->scoreMin.init(SIMPLE_FUNC_LINEAR, -0.6f, -0.6f);
	c->nCeil.init(SIMPLE_FUNC_LINEAR, 0.0f, std::numeric_limits<double>::max(), 0.0f, 0.15f);
	return new Scoring(
		0, COST_MODEL_QUAL, 6, 2, c->scoreMin, c->nCeil,
		COST_MODEL_CONSTANT, 1, false, 5, 5, 3, 3, 4);
}

extern "C" {

void *ref_open(const char *base) {
	try {
		RefCtx *c = new RefCtx();
		std::string b(base);
		c->fw = new Ebwt(b, 0, -1, true, -1, 0, false, false, false,
		                 true, true, true, true, false, false, false, false);
		c->bw = new Ebwt(b + ".rev", 0, 1, false, -1, 0, false, false, false,
		                 true, true, true, true, false, false, false, false);
		c->fw->loadIntoMemory(0, -1, true, true, true, true, false);
		c->bw->loadIntoMemory(0, 1, false, true, false, true, false);
		c->ref = new BitPairReference(b, false, false, NULL, NULL, false, false, false, false, false, false);
		if(!c->ref->loaded()) return NULL;
		c->sc = make_default_scoring(c);
		return c;
	} catch(...) {
		return NULL;
	}
}

void ref_close(void *h) {
	RefCtx *c = (RefCtx*)h;
	delete c->fw; delete c->bw; delete c->ref; delete c->sc; delete c;
}

int ref_off_size() { return (int)OFF_SIZE; }

static const Ebwt& E(void *h, int dir) { RefCtx *c = (RefCtx*)h; return dir == 0 ? *c->fw : *c->bw; }

uint64_t ref_len(void *h) { return E(h, 0).eh().len(); }
uint64_t ref_zoff(void *h, int dir) { return E(h, dir).zOff(); }

/* countBt2SideEx through a SideLocus, exactly as mapBiLFEx does (bt2_idx.h:2372) */
void ref_rank4(void *h, int dir, uint64_t row, uint64_t *out) {
	const Ebwt& e = E(h, dir);
	SideLocus l;
	l.initFromRow((TIndexOffU)row, e.eh(), e.ebwt());
	TIndexOffU a[4] = {0, 0, 0, 0};
	e.countBt2SideEx(l, a);
	for(int i = 0; i < 4; i++) out[i] = a[i];
}

uint64_t ref_rank(void *h, int dir, uint64_t row, int c) {
	const Ebwt& e = E(h, dir);
	SideLocus l;
	l.initFromRow((TIndexOffU)row, e.eh(), e.ebwt());
	return e.mapLF(l, c);
}

uint64_t ref_map_lf1c(void *h, int dir, uint64_t row, int c) {
	const Ebwt& e = E(h, dir);
	SideLocus l;
	l.initFromRow((TIndexOffU)row, e.eh(), e.ebwt());
	TIndexOffU r = e.mapLF1((TIndexOffU)row, l, c);
	return r == (TIndexOffU)OFF_MASK ? UINT64_MAX : (uint64_t)r;
}

int ref_row_l(void *h, int dir, uint64_t row) { return E(h, dir).rowL((TIndexOffU)row); }

void ref_ftab_lohi(void *h, int dir, uint64_t key, uint64_t *top, uint64_t *bot) {
	TIndexOffU t, b;
	E(h, dir).ftabLoHi((TIndexOffU)key, t, b);
	*top = t; *bot = b;
}

uint64_t ref_get_offset(void *h, uint64_t row) { return E(h, 0).getOffset((TIndexOffU)row); }

void ref_joined_to_text_off(void *h, uint64_t qlen, uint64_t off, uint64_t *tidx,
                            uint64_t *textoff, uint64_t *tlen, int reject, int *straddled) {
	TIndexOffU ti = 0, to = 0, tl = 0; bool st = false;
	E(h, 0).joinedToTextOff((TIndexOffU)qlen, (TIndexOffU)off, ti, to, tl, reject != 0, st);
	*tidx = (ti == (TIndexOffU)OFF_MASK) ? UINT64_MAX : (uint64_t)ti;
	*textoff = to; *tlen = tl; *straddled = st ? 1 : 0;
}

int ref_get_base(void *h, uint64_t tidx, uint64_t toff) { return ((RefCtx*)h)->ref->getBase((size_t)tidx, (size_t)toff); }

/* SeedAligner::exactSweep on an ASCII read; out = {mineFw,mineRc,hitFw,hitRc,topFw,botFw,topRc,botRc,nelt,bwops} */
void ref_exact_sweep(void *h, const char *seq, const char *qual, uint64_t *out) {
	RefCtx *c = (RefCtx*)h;
	Read rd("r", seq, qual);
	SeedAligner al;
	SeedResults shs;
	SeedSearchMetrics met;
	shs.nextRead(rd);
	size_t mfw = 0, mrc = 0;
	al.bwops_ = 0;
	size_t nelt = al.exactSweep(*c->fw, rd, *c->sc, false, false, 2, mfw, mrc, true, shs, met);
	EEHit f = shs.exactFwEEHit(), r = shs.exactRcEEHit();
	out[0] = mfw; out[1] = mrc;
	out[2] = f.empty() ? 0 : 1; out[3] = r.empty() ? 0 : 1;
	out[4] = f.empty() ? 0 : f.top; out[5] = f.empty() ? 0 : f.bot;
	out[6] = r.empty() ? 0 : r.top; out[7] = r.empty() ? 0 : r.bot;
	out[8] = nelt; out[9] = al.bwops_;
}

/* SeedAligner::oneMmSearch on an ASCII read.  Per hit, in the order the search added them: {top, bot, score, pos (from the 5' end), reference
 * character code, read character code, fw}.  Returns the number of 1-mismatch hits (at most cap are written); out_exact = {fw hit?, top,
 * bot, rc hit?, top, bot} for repex. */
int ref_one_mm(void *h, const char *seq, const char *qual, int64_t minsc, int nofw, int norc, int local, int repex, int rep1mm,
               uint64_t *out, int cap, uint64_t *out_exact) {
	RefCtx *c = (RefCtx*)h;
	Read rd("r", seq, qual);
	SeedAligner al;
	SeedResults shs;
	SeedSearchMetrics met;
	shs.nextRead(rd);
	al.oneMmSearch(c->fw, c->bw, rd, *c->sc, minsc, nofw != 0, norc != 0, local != 0, repex != 0, rep1mm != 0, shs, met);
	const EList<EEHit>& hs = shs.mm1EEHits();
	auto code = [](int ch) -> uint64_t { switch(ch) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; } };
	for(size_t i = 0; i < hs.size() && (int)i < cap; i++) {
		uint64_t *o = out + i * 7;
		o[0] = hs[i].top; o[1] = hs[i].bot; o[2] = (uint64_t)hs[i].score; o[3] = hs[i].e1.pos; o[4] = code(hs[i].e1.chr); o[5] = code(hs[i].e1.qchr); o[6] = hs[i].fw ? 1 : 0;
	}
	EEHit f = shs.exactFwEEHit(), r = shs.exactRcEEHit();
	out_exact[0] = f.empty() ? 0 : 1; out_exact[1] = f.empty() ? 0 : f.top; out_exact[2] = f.empty() ? 0 : f.bot;
	out_exact[3] = r.empty() ? 0 : 1; out_exact[4] = r.empty() ? 0 : r.top; out_exact[5] = r.empty() ? 0 : r.bot;
	return (int)hs.size();
}

/*
 * One exact seeding round as the worker runs it (bt2_search.cpp:3910-3964):
 * Seed::mmSeeds(0, seedlen) -> instantiateSeeds(offset, interval) -> searchAllSeeds.
 * Writes, per (fw?0:1, offidx): present, topf, botf, topb, botb (5 u64 each;
 * stride 5, index = fwi*nseeds + i).  Returns nseeds (numOffs).
 */
int ref_seed_round(void *h, const char *seq, const char *qual, int seedlen, int interval,
                   int offset, uint64_t *out, int out_cap, uint64_t *bwops) {
	RefCtx *c = (RefCtx*)h;
	Read rd("r", seq, qual);
	SeedAligner al;
	SeedResults shs;
	SeedSearchMetrics met;
	PerReadMetrics prm;
	AlignmentCache scCurrent(20 * 1024 * 1024, false);
	AlignmentCacheIface ca(&scCurrent, NULL, NULL);
	ca.nextRead();
	shs.nextRead(rd);
	EList<Seed> seeds;
	Constraint gc = Constraint::penaltyFuncBased(c->scoreMin);
	Seed::mmSeeds(0, seedlen, seeds, gc);
	std::pair<int, int> instFw, instRc;
	std::pair<int, int> inst = al.instantiateSeeds(seeds, offset, interval, rd, *c->sc, false, false,
	                                               ca, shs, met, instFw, instRc);
	int n = (int)shs.numOffs();
	if(n * 2 * 5 > out_cap) return -1;
	memset(out, 0, sizeof(uint64_t) * n * 2 * 5);
	if(inst.first + inst.second == 0) { *bwops = 0; return n; }
	al.searchAllSeeds(seeds, c->fw, c->bw, rd, *c->sc, std::numeric_limits<size_t>::max(), ca, shs, met, prm);
	*bwops = met.bwops;
	EList<SATuple, 16> satups;
	for(int fwi = 0; fwi < 2; fwi++) {
		for(int i = 0; i < n; i++) {
			const QVal& qv = shs.hitsAtOffIdx(fwi == 0, i);
			uint64_t *o = out + (fwi * n + i) * 5;
			if(!qv.valid() || qv.empty()) continue;
			satups.clear();
			size_t nrange = 0, nelt = 0;
			ca.queryQval(qv, satups, nrange, nelt);
			if(satups.size() != 1) { o[0] = 100 + satups.size(); continue; }
			o[0] = 1;
			o[1] = satups[0].topf; o[2] = satups[0].topf + satups[0].size();
			o[3] = satups[0].topb; o[4] = satups[0].topb + satups[0].size();
		}
	}
	return n;
}

/*
 * Reference end-to-end u8 fill (alignNucleotidesEnd2EndSseU8) on an explicit
 * problem: read chars (ASCII ACGTN) + ASCII quals in alignment orientation,
 * reference masks rf[cols+1] (one extra column as initRef captures).  Copies the
 * filled H/E/F (u8 encoded) to row-major [rows*cols] buffers; returns best score.
 */
int64_t ref_sw_fill_ee_u8(void *h, const char *seq, const char *qual, const uint8_t *rf, int cols,
                          int64_t minsc, uint8_t *H, uint8_t *Eo, uint8_t *F, int *flag_out) {
	RefCtx *c = (RefCtx*)h;
	Read rd("r", seq, qual);
	SwAligner sw(NULL);
	size_t rows = rd.length();
	sw.initRead(rd.patFw, rd.patRc, rd.qual, rd.qualRev, 0, rows, *c->sc);
	DPRect rect;
	rect.refl = rect.refl_pretrim = 0;
	rect.refr = rect.refr_pretrim = cols - 1;
	rect.triml = rect.trimr = 0;
	rect.corel = 0; rect.corer = cols - 1; rect.maxgap = 15;
	std::vector<char> rfbuf(rf, rf + cols + 1);
	sw.initRef(true, 0, rect, rfbuf.data(), 0, (size_t)cols, 1000000, *c->sc, minsc,
	           true, 2000, 4, true, true);
	int flag = 0;
	sw.sse8succ_ = false;
	int64_t best = sw.alignNucleotidesEnd2EndSseU8(flag, false);
	if(flag_out) *flag_out = flag;
	SSEData& d = sw.sseU8fw_;
	for(size_t i = 0; i < rows; i++) {
		for(int j = 0; j < cols; j++) {
			H[i * cols + j]  = (uint8_t)d.mat_.helt(i, j);
			Eo[i * cols + j] = (uint8_t)d.mat_.eelt(i, j);
			F[i * cols + j]  = (uint8_t)d.mat_.felt(i, j);
		}
	}
	return best;
}

/*
 * The other three fills of the path on the same explicit problem: end-to-end 16-bit (alignNucleotidesEnd2EndSseI16,
 * aligner_swsse_ee_i16.cpp:780), local 8-bit (alignNucleotidesLocalSseU8, aligner_swsse_loc_u8.cpp:927) and local 16-bit
 * (alignNucleotidesLocalSseI16, aligner_swsse_loc_i16.cpp:938).  kind: 1 = ee i16, 2 = local u8, 3 = local i16.  H/E/F come back as
 * the matrix holds them (SSEMatrix::elt: u8 or i16 words, biased as the kernel stores them), row-major int32; returns the kernel's score.
 */
int64_t ref_sw_fill_kind(void *h, int kind, const char *seq, const char *qual, const uint8_t *rf, int cols,
                         int64_t minsc, int32_t *H, int32_t *Eo, int32_t *F, int *flag_out) {
	RefCtx *c = (RefCtx*)h;
	Read rd("r", seq, qual);
	SwAligner sw(NULL);
	size_t rows = rd.length();
	sw.initRead(rd.patFw, rd.patRc, rd.qual, rd.qualRev, 0, rows, *c->sc);
	DPRect rect;
	rect.refl = rect.refl_pretrim = 0;
	rect.refr = rect.refr_pretrim = cols - 1;
	rect.triml = rect.trimr = 0;
	rect.corel = 0; rect.corer = cols - 1; rect.maxgap = 15;
	std::vector<char> rfbuf(rf, rf + cols + 1);
	sw.initRef(true, 0, rect, rfbuf.data(), 0, (size_t)cols, 1000000, *c->sc, minsc,
	           true, 2000, 4, true, true);
	int flag = 0;
	sw.sse8succ_ = false;
	sw.sse16succ_ = false;
	int64_t best = kind == 1 ? sw.alignNucleotidesEnd2EndSseI16(flag, false)
	             : kind == 2 ? sw.alignNucleotidesLocalSseU8(flag, false)
	                         : sw.alignNucleotidesLocalSseI16(flag, false);
	if(flag_out) *flag_out = flag;
	SSEData& d = kind == 2 ? sw.sseU8fw_ : sw.sseI16fw_;
	for(size_t i = 0; i < rows; i++) {
		for(int j = 0; j < cols; j++) {
			H[i * cols + j]  = d.mat_.helt(i, j);
			Eo[i * cols + j] = d.mat_.eelt(i, j);
			F[i * cols + j]  = d.mat_.felt(i, j);
		}
	}
	return best;
}

/* the shim's scoring scheme switched to local mode and back (match bonus; SwAligner reads monotone_ off the scheme) */
void ref_set_match_bonus(void *h, int bonus) {
	RefCtx *c = (RefCtx*)h;
	c->sc->setMatchBonus(bonus);
	c->sc->monotone = (bonus == 0);      // Scoring's constructor derives it once (scoring.h:168)
}

/* RandomSource stream check: fills out[n] following ops[i]: 0=nextU32 1=nextBool 2=nextU2 3=nextFloat(bits) 4=nextU64(lo32^hi32) */
void ref_rng_stream(uint32_t seed, const uint8_t *ops, int n, uint32_t *out) {
	RandomSource r;
	r.init(seed);
	for(int i = 0; i < n; i++) {
		switch(ops[i]) {
			case 0: out[i] = r.nextU32(); break;
			case 1: out[i] = r.nextBool() ? 1 : 0; break;
			case 2: out[i] = r.nextU2(); break;
			case 3: { float f = r.nextFloat(); memcpy(&out[i], &f, 4); break; }
			default: { uint64_t v = r.nextU64(); out[i] = (uint32_t)(v ^ (v >> 32)); break; }
		}
	}
}

int64_t ref_score(void *h, int rdc, int refm, int q) { return ((RefCtx*)h)->sc->score(rdc, refm, q); }

} // extern "C"
