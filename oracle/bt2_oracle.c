/*
 * oracle/bt2_oracle.c -- TEST INFRASTRUCTURE ONLY (see bt2_oracle.h).
 *
 * Scalar plain-C restatement of the reference's FM-index rank/LF, exact
 * sweep, exact seed search, offset resolution, reference fetch, RNG and the
 * four DP fills (end-to-end 8/16-bit, local 8/16-bit) as their fixed points in
 * saturating arithmetic.  Written from the behaviour documented in
 * SURVEY.md Appendix A-C; every function names the reference lines it follows.
 * No code is copied: the reference's byte-LUT + SSE formulation is replaced by
 * straight loops over 2-bit characters.
 */
#include "bt2_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* file helpers                                                        */
/* ------------------------------------------------------------------ */
static int rd_bytes(FILE *f, void *dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

static int rd_off(FILE *f, int off_size, uint64_t *v) {
	if (off_size == 4) { uint32_t x; if (rd_bytes(f, &x, 4)) return -1; *v = x; return 0; }
	return rd_bytes(f, v, 8);
}

static int rd_off_arr(FILE *f, int off_size, uint64_t n, uint64_t **out) {
	uint64_t *a = (uint64_t *)malloc((size_t)(n ? n : 1) * 8);
	if (!a) return -1;
	if (off_size == 8) {
		if (rd_bytes(f, a, (size_t)n * 8)) { free(a); return -1; }
	} else {
		uint32_t *t = (uint32_t *)malloc((size_t)(n ? n : 1) * 4);
		if (!t || rd_bytes(f, t, (size_t)n * 4)) { free(t); free(a); return -1; }
		for (uint64_t i = 0; i < n; i++) a[i] = t[i];
		free(t);
	}
	*out = a;
	return 0;
}

static int file_exists(const char *p) { FILE *f = fopen(p, "rb"); if (f) { fclose(f); return 1; } return 0; }

/* Ebwt::readIntoMemory, bt2_io.cpp:39-633; header layout SURVEY.md Appendix B. */
static int load_ebwt(bt2o_ebwt *e, const char *p1, const char *p2, int off_size, int fw,
                     int load_sa, int load_rstarts, int load_names) {
	memset(e, 0, sizeof(*e));
	e->off_size = off_size;
	e->off_mask = off_size == 4 ? 0xffffffffull : ~0ull;
	e->fw = fw;
	FILE *f = fopen(p1, "rb");
	if (!f) return -1;
	int32_t one, lps;
	if (rd_bytes(f, &one, 4) || one != 1) { fclose(f); return -2; } /* little-endian only */
	if (rd_off(f, off_size, &e->len)) goto bad;
	if (rd_bytes(f, &e->line_rate, 4) || rd_bytes(f, &lps, 4) || rd_bytes(f, &e->off_rate, 4) ||
	    rd_bytes(f, &e->ftab_chars, 4) || rd_bytes(f, &e->flags, 4)) goto bad;
	/* EbwtParams::init, bt2_idx.h:133-167 */
	e->side_sz = 1u << e->line_rate;
	e->side_bwt_sz = e->side_sz - 4 * (uint32_t)off_size;
	e->side_bwt_len = e->side_bwt_sz * 4;
	if (e->side_bwt_len != (uint32_t)(48 * off_size)) goto bad; /* reference hard-codes 48*OFF_SIZE, bt2_idx.h:372 */
	{
		uint64_t bwt_sz = e->len / 4 + 1;
		e->num_sides = (bwt_sz + e->side_bwt_sz - 1) / e->side_bwt_sz;
		e->ebwt_tot_len = e->num_sides * e->side_sz;
		e->ftab_len = (1ull << (2 * e->ftab_chars)) + 1;
		e->eftab_len = (uint64_t)e->ftab_chars * 2;
		e->offs_len = (e->len + 1 + (1ull << e->off_rate) - 1) >> e->off_rate;
	}
	if (rd_off(f, off_size, &e->n_pat)) goto bad;
	if (rd_off_arr(f, off_size, e->n_pat, &e->plen)) goto bad;
	if (rd_off(f, off_size, &e->n_frag)) goto bad;
	if (load_rstarts) {
		if (rd_off_arr(f, off_size, e->n_frag * 3, &e->rstarts)) goto bad;
	} else {
		if (fseeko(f, (off_t)(e->n_frag * 3 * off_size), SEEK_CUR)) goto bad;
	}
	e->ebwt = (uint8_t *)malloc((size_t)e->ebwt_tot_len);
	if (!e->ebwt || rd_bytes(f, e->ebwt, (size_t)e->ebwt_tot_len)) goto bad;
	if (rd_off(f, off_size, &e->zoff)) goto bad;
	for (int i = 0; i < 5; i++) if (rd_off(f, off_size, &e->fchr[i])) goto bad;
	if (rd_off_arr(f, off_size, e->ftab_len, &e->ftab)) goto bad;
	if (rd_off_arr(f, off_size, e->eftab_len, &e->eftab)) goto bad;
	if (load_names) {
		/* '\n'-separated names ending in '\0' (bt2_io.cpp:468-484) */
		size_t cap = 16, n = 0;
		e->refnames = (char **)calloc(cap, sizeof(char *));
		char buf[4096]; size_t bl = 0; int any = 0;
		for (;;) {
			int c = fgetc(f);
			if (c == EOF || c == '\0' || c == '\n') {
				if (any || bl > 0 || c == '\n') {
					if (n == cap) { cap *= 2; e->refnames = (char **)realloc(e->refnames, cap * sizeof(char *)); }
					buf[bl] = 0;
					e->refnames[n++] = strdup(buf);
					bl = 0; any = 0;
				}
				if (c != '\n') break;
			} else {
				if (bl + 1 < sizeof(buf)) buf[bl++] = (char)c;
				any = 1;
			}
		}
		e->n_refnames = n;
	}
	fclose(f);
	if (load_sa) {
		FILE *g = fopen(p2, "rb");
		if (!g) return -3;
		if (rd_bytes(g, &one, 4) || one != 1) { fclose(g); return -3; }
		if (rd_off_arr(g, off_size, e->offs_len, &e->offs)) { fclose(g); return -3; }
		fclose(g);
	}
	return 0;
bad:
	fclose(f);
	return -4;
}

/* BitPairReference::BitPairReference, reference.cpp:30-264 */
static int load_ref(bt2o_ref *r, const char *p3, const char *p4, int off_size) {
	memset(r, 0, sizeof(*r));
	FILE *f = fopen(p3, "rb");
	if (!f) return -1;
	int32_t one;
	if (rd_bytes(f, &one, 4) || one != 1) { fclose(f); return -2; }
	if (rd_off(f, off_size, &r->nrecs) || r->nrecs == 0) { fclose(f); return -2; }
	r->rec_off = (uint64_t *)malloc(r->nrecs * 8);
	r->rec_len = (uint64_t *)malloc(r->nrecs * 8);
	r->rec_first = (uint8_t *)malloc(r->nrecs);
	r->ref_rec_offs = (uint64_t *)malloc((r->nrecs + 1) * 8);
	r->ref_offs = (uint64_t *)malloc((r->nrecs + 1) * 8);
	r->ref_lens = (uint64_t *)malloc((r->nrecs + 1) * 8);
	uint64_t cumsz = 0, cumlen = 0;
	for (uint64_t i = 0; i < r->nrecs; i++) {
		int c;
		if (rd_off(f, off_size, &r->rec_off[i]) || rd_off(f, off_size, &r->rec_len[i]) || (c = fgetc(f)) == EOF) {
			fclose(f); return -2;
		}
		r->rec_first[i] = c ? 1 : 0;
		if (r->rec_first[i]) {
			r->ref_rec_offs[r->nrefs] = i;
			r->ref_offs[r->nrefs] = cumsz;
			if (r->nrefs > 0) r->ref_lens[r->nrefs - 1] = cumlen;
			cumlen = 0;
			r->nrefs++;
		} else if (i == 0) { fclose(f); return -2; }
		cumsz += r->rec_len[i];
		cumlen += r->rec_off[i] + r->rec_len[i];
	}
	fclose(f);
	r->ref_rec_offs[r->nrefs] = r->nrecs;
	r->ref_offs[r->nrefs] = cumsz;
	r->ref_lens[r->nrefs - 1] = cumlen;
	r->buf_sz = cumsz;
	uint64_t nbytes = (cumsz + 3) / 4;
	r->buf = (uint8_t *)malloc((size_t)(nbytes ? nbytes : 1));
	FILE *g = fopen(p4, "rb");
	if (!g) return -3;
	if (rd_bytes(g, r->buf, (size_t)nbytes)) { fclose(g); return -3; }
	fclose(g);
	return 0;
}

int bt2o_index_load(bt2o_index *idx, const char *base) {
	char p1[4096], p2[4096], p3[4096], p4[4096];
	memset(idx, 0, sizeof(*idx));
	const char *ext = "bt2"; int off_size = 4;
	snprintf(p1, sizeof p1, "%s.1.bt2", base);
	if (!file_exists(p1)) {
		snprintf(p1, sizeof p1, "%s.1.bt2l", base);
		if (!file_exists(p1)) return -1;
		ext = "bt2l"; off_size = 8;
	}
	snprintf(p1, sizeof p1, "%s.1.%s", base, ext);
	snprintf(p2, sizeof p2, "%s.2.%s", base, ext);
	int rc = load_ebwt(&idx->fwd, p1, p2, off_size, 1, 1, 1, 1);
	if (rc) return rc;
	snprintf(p1, sizeof p1, "%s.rev.1.%s", base, ext);
	if (file_exists(p1)) {
		rc = load_ebwt(&idx->bwd, p1, NULL, off_size, 0, 0, 0, 0);
		if (rc) return rc;
		idx->has_bwd = 1;
	}
	snprintf(p3, sizeof p3, "%s.3.%s", base, ext);
	snprintf(p4, sizeof p4, "%s.4.%s", base, ext);
	if (file_exists(p3) && file_exists(p4)) {
		rc = load_ref(&idx->ref, p3, p4, off_size);
		if (rc) return rc;
		idx->has_ref = 1;
	}
	return 0;
}

static void free_ebwt(bt2o_ebwt *e) {
	free(e->plen); free(e->rstarts); free(e->ebwt); free(e->ftab); free(e->eftab); free(e->offs);
	if (e->refnames) { for (size_t i = 0; i < e->n_refnames; i++) free(e->refnames[i]); free(e->refnames); }
	memset(e, 0, sizeof(*e));
}

void bt2o_index_free(bt2o_index *idx) {
	free_ebwt(&idx->fwd);
	if (idx->has_bwd) free_ebwt(&idx->bwd);
	if (idx->has_ref) {
		bt2o_ref *r = &idx->ref;
		free(r->rec_off); free(r->rec_len); free(r->rec_first); free(r->ref_rec_offs);
		free(r->ref_offs); free(r->ref_lens); free(r->buf);
	}
	memset(idx, 0, sizeof(*idx));
}

/* ------------------------------------------------------------------ */
/* rank / LF                                                           */
/* ------------------------------------------------------------------ */
static inline uint64_t side_occ(const bt2o_ebwt *e, const uint8_t *side, int c) {
	const uint8_t *p = side + e->side_bwt_sz + (size_t)c * e->off_size;
	if (e->off_size == 4) { uint32_t v; memcpy(&v, p, 4); return v; }
	uint64_t v; memcpy(&v, p, 8); return v;
}

static inline int bwt_char(const uint8_t *side, uint32_t i) { return (side[i >> 2] >> ((i & 3) * 2)) & 3; }

/* countBt2SideEx (bt2_idx.h:1887): counts in [0,charOff) of the side, '$' fix, + occ + fchr */
void bt2o_rank4(const bt2o_ebwt *e, uint64_t row, uint64_t out[4]) {
	uint64_t side_num = row / e->side_bwt_len;
	uint32_t char_off = (uint32_t)(row % e->side_bwt_len);
	const uint8_t *side = e->ebwt + side_num * e->side_sz;
	uint64_t cnt[4] = {0, 0, 0, 0};
	for (uint32_t i = 0; i < char_off; i++) cnt[bwt_char(side, i)]++;
	/* '$' is stored as 'A' but must not count (bt2_idx.h:1891-1899) */
	if (side_num == e->zoff / e->side_bwt_len && char_off > (uint32_t)(e->zoff % e->side_bwt_len)) cnt[0]--;
	for (int c = 0; c < 4; c++) out[c] = cnt[c] + side_occ(e, side, c) + e->fchr[c];
}

uint64_t bt2o_rank(const bt2o_ebwt *e, uint64_t row, int c) {
	uint64_t r[4];
	bt2o_rank4(e, row, r);
	return r[c];
}

int bt2o_row_l(const bt2o_ebwt *e, uint64_t row) {
	uint64_t side_num = row / e->side_bwt_len;
	uint32_t char_off = (uint32_t)(row % e->side_bwt_len);
	return bwt_char(e->ebwt + side_num * e->side_sz, char_off);
}

uint64_t bt2o_map_lf(const bt2o_ebwt *e, uint64_t row) { return bt2o_rank(e, row, bt2o_row_l(e, row)); }

uint64_t bt2o_map_lf1c(const bt2o_ebwt *e, uint64_t row, int c) {
	if (bt2o_row_l(e, row) != c || row == e->zoff) return e->off_mask;
	return bt2o_rank(e, row, c);
}

int bt2o_map_lf1(const bt2o_ebwt *e, uint64_t *row) {
	if (*row == e->zoff) return -1;
	int c = bt2o_row_l(e, *row);
	*row = bt2o_rank(e, *row, c);
	return c;
}

/* ftabHi/ftabLo with eftab indirection, bt2_idx.h:1428-1554 */
static uint64_t ftab_hi(const bt2o_ebwt *e, uint64_t i) {
	if (e->ftab[i] <= e->len) return e->ftab[i];
	uint64_t ef = (e->ftab[i] ^ e->off_mask);
	return e->eftab[ef * 2 + 1];
}
static uint64_t ftab_lo(const bt2o_ebwt *e, uint64_t i) {
	if (e->ftab[i] <= e->len) return e->ftab[i];
	uint64_t ef = (e->ftab[i] ^ e->off_mask);
	return e->eftab[ef * 2];
}
void bt2o_ftab_lohi(const bt2o_ebwt *e, uint64_t key, uint64_t *top, uint64_t *bot) {
	*top = ftab_hi(e, key);
	*bot = ftab_lo(e, key + 1);
}

uint64_t bt2o_ftab_seq_to_int(const bt2o_ebwt *e, const uint8_t *seq, size_t off, int rev) {
	int fc = e->ftab_chars;
	size_t lo = off, hi = off + (size_t)fc;
	uint64_t k = 0;
	int fwex = e->fw ? 1 : 0;
	if (rev) fwex = !fwex;
	for (int i = 0; i < fc; i++) {
		int c = fwex ? seq[lo + i] : seq[hi - i - 1];
		if (c > 3) return UINT64_MAX;
		k = (k << 2) | (uint64_t)c;
	}
	return k;
}

uint64_t bt2o_get_offset(const bt2o_ebwt *e, uint64_t row, uint64_t *nsteps) {
	uint64_t jumps = 0;
	uint64_t mask = (e->off_mask << e->off_rate) & e->off_mask;
	for (;;) {
		if (row == e->zoff) { if (nsteps) *nsteps = jumps; return jumps; }
		if ((row & mask) == row) { if (nsteps) *nsteps = jumps; return jumps + e->offs[row >> e->off_rate]; }
		row = bt2o_map_lf(e, row);
		jumps++;
	}
}

void bt2o_joined_to_text_off(const bt2o_ebwt *e, uint64_t qlen, uint64_t off,
                             uint64_t *tidx, uint64_t *textoff, uint64_t *tlen,
                             int reject_straddle, int *straddled) {
	uint64_t top = 0, bot = e->n_frag;
	*straddled = 0;
	*tidx = e->off_mask; *textoff = 0; *tlen = 0;
	if (off >= e->len || e->n_frag == 0) return;   /* outside the joined text: the reference never asks */
	for (;;) {
		uint64_t elt = top + ((bot - top) >> 1);
		uint64_t lower = e->rstarts[elt * 3];
		uint64_t upper = (elt == e->n_frag - 1) ? e->len : e->rstarts[(elt + 1) * 3];
		uint64_t fraglen = upper - lower;
		if (lower <= off) {
			if (upper > off) {
				if (off + qlen > upper) {
					*straddled = 1;
					if (reject_straddle) { *tidx = e->off_mask; return; }
				}
				*tidx = e->rstarts[elt * 3 + 1];
				uint64_t fragoff = off - lower;
				if (!e->fw) { fragoff = fraglen - fragoff - 1; fragoff -= (qlen - 1); }
				*textoff = fragoff + e->rstarts[elt * 3 + 2];
				break;
			}
			top = elt;
		} else {
			bot = elt;
		}
	}
	*tlen = e->plen[*tidx];
}

/* ------------------------------------------------------------------ */
/* exact sweep                                                         */
/* ------------------------------------------------------------------ */
/* One LF step on a (top,bot) pair for char c as exactSweepMapLF does
 * (aligner_seed.cpp:793-824): 2 rank queries when bot-top>1, one when ==1. */
static void pair_lf(const bt2o_ebwt *e, int c, uint64_t *top, uint64_t *bot, uint64_t *bwops, uint64_t *nrank) {
	if (c > 3) { *top = *bot = 0; return; }
	if (*bot - *top > 1) {
		*bwops += 2;
		/* SideLocus::initFromTopBot: one side read if both loci share a side (bt2_idx.h:339-348) */
		*nrank += ((*top / e->side_bwt_len) == (*bot / e->side_bwt_len)) ? 1 : 2;
		*top = bt2o_rank(e, *top, c);
		*bot = bt2o_rank(e, *bot, c);
	} else {
		*bwops += 1; *nrank += 1;
		uint64_t t = bt2o_map_lf1c(e, *top, c);
		if (t == e->off_mask) { *top = *bot = 0; }
		else { *top = t; *bot = t + 1; }
	}
}

void bt2o_exact_sweep(const bt2o_ebwt *e, const uint8_t *seq_fw, const uint8_t *seq_rc,
                      size_t len, int nofw, int norc, uint32_t mine_max, bt2o_sweep_out *out) {
	memset(out, 0, sizeof(*out));
	const int ftab_len = e->ftab_chars;
	/* The reference interleaves fw and rc only to overlap prefetches
	 * (aligner_seed.cpp:913); the two strands are independent, so do them in turn. */
	for (int fwi = 0; fwi < 2; fwi++) {
		if ((fwi == 0 && nofw) || (fwi == 1 && norc)) continue;
		const uint8_t *seq = fwi == 0 ? seq_fw : seq_rc;
		size_t dep = 0; uint32_t nedit = 0; int done = 0, do_init = 1;
		uint64_t top = 0, bot = 0;
		while (dep < len && !done) {
			if (do_init) {
				/* exactSweepInit :752-791 */
				top = bot = 0;
				size_t left = len - dep;
				int do_ftab = ftab_len > 1 && left >= (size_t)ftab_len;
				if (do_ftab) {
					size_t endi = len - dep - 1;
					for (int i = 0; i < ftab_len; i++) if (seq[endi - i] > 3) { do_ftab = 0; break; }
				}
				if (do_ftab) {
					uint64_t key = bt2o_ftab_seq_to_int(e, seq, left - ftab_len, 0);
					bt2o_ftab_lohi(e, key, &top, &bot);
					dep += ftab_len;
				} else {
					int c = seq[len - dep - 1];
					if (c < 4) { top = e->fchr[c]; bot = e->fchr[c + 1]; }
					dep++;
				}
				/* exactSweepStep :826-848 */
				if (bot <= top) {
					nedit++;
					if (nedit >= mine_max) { out->mine[fwi] = nedit; done = 1; }
					continue;
				}
				do_init = 0;
			}
			if (dep < len) {
				pair_lf(e, seq[len - dep - 1], &top, &bot, &out->bwops, &out->nrank);
				if (bot <= top) {
					nedit++;
					if (nedit >= mine_max) { out->mine[fwi] = nedit; done = 1; }
					do_init = 1;
				}
				dep++;
			}
		}
		if (!done && dep >= len) {
			out->mine[fwi] = nedit;
			if (nedit == 0 && bot > top) {
				out->hit[fwi] = 1;
				out->top[fwi] = top; out->bot[fwi] = bot;
				out->nelt += bot - top;
			}
		}
	}
}

/* ------------------------------------------------------------------ */
/* exact seed search                                                   */
/* ------------------------------------------------------------------ */
void bt2o_seed_search_exact(const bt2o_ebwt *fw, const bt2o_ebwt *bw,
                            const uint8_t *seq, size_t len, bt2o_seed_hit *out) {
	memset(out, 0, sizeof(*out));
	const int ftab_len = fw->ftab_chars;
	uint64_t topf, botf, topb, botb;
	size_t step = 0;
	/* steps[k] = -(len-k): right-to-left over the forward index (aligner_seed.cpp:256-262).
	 * startSearchSeedBi :1638-1718: ftab jump when ftabChars <= maxjump (= len, no Ns). */
	if (ftab_len > 1 && (size_t)ftab_len <= len) {
		size_t off = len - ftab_len;
		uint64_t fwi0 = bt2o_ftab_seq_to_int(fw, seq, off, 0);
		bt2o_ftab_lohi(fw, fwi0, &topf, &botf);
		if (botf - topf == 0) return;
		uint64_t bwi0 = bt2o_ftab_seq_to_int(bw, seq, off, 0);
		topb = ftab_hi(bw, bwi0);            /* NDEBUG branch :1678-1682 */
		botb = topb + (botf - topf);
		step = ftab_len;
	} else {
		int c = seq[len - 1];
		topf = topb = fw->fchr[c];
		botf = botb = fw->fchr[c + 1];
		if (botf - topf == 0) return;
		step = 1;
	}
	for (; step < len; step++) {
		int c = seq[len - step - 1];
		if (botf - topf > 1) {
			/* mapBiLFEx (bt2_idx.h:2372): both ranks for all four chars + prefix sums in BWT' */
			uint64_t t[4], b[4];
			out->bwops++;
			out->nrank += ((topf / fw->side_bwt_len) == (botf / fw->side_bwt_len)) ? 1 : 2;
			bt2o_rank4(fw, topf, t);
			bt2o_rank4(fw, botf, b);
			uint64_t tp = topb;
			for (int j = 0; j < c; j++) tp += b[j] - t[j];
			if (b[c] == t[c]) { out->topf = out->botf = out->topb = out->botb = 0; return; }
			topf = t[c]; botf = b[c];
			topb = tp; botb = tp + (b[c] - t[c]);
		} else {
			out->bwops++; out->nrank++;
			uint64_t t = bt2o_map_lf1c(fw, topf, c);
			if (t == fw->off_mask) { out->topf = out->botf = out->topb = out->botb = 0; return; }
			topf = t; botf = t + 1; /* topb/botb unchanged (:2003-2016) */
		}
	}
	out->topf = topf; out->botf = botf; out->topb = topb; out->botb = botb;
}

/* ------------------------------------------------------------------ */
/* 1-mismatch end-to-end search                                        */
/* ------------------------------------------------------------------ */
/* Ebwt::mapBiLFEx (bt2_idx.h:2372-2440): ranks of all four characters at top and bot, and the sub-ranges in the other index
 * (prefix sums in character order from topp). */
static void bi_lf_ex(const bt2o_ebwt *e, uint64_t top, uint64_t bot, uint64_t topp, uint64_t t[4], uint64_t b[4], uint64_t tp[4], uint64_t bp[4]) {
	bt2o_rank4(e, top, t);
	bt2o_rank4(e, bot, b);
	uint64_t acc = topp;
	for (int j = 0; j < 4; j++) { tp[j] = acc; acc += b[j] - t[j]; bp[j] = acc; }
}

int bt2o_one_mm_search(const bt2o_ebwt *ebwt_fw, const bt2o_ebwt *ebwt_bw, const uint8_t *seq_in, const char *qual_in, size_t len,
                       const bt2o_scoring *sc, int nceil, int64_t minsc, int nofw, int norc, int local, int repex, int rep1mm,
                       bt2o_mm1_hit *out, int cap) {
	const int match_bonus = sc->match_bonus;
	int nout = 0;
	size_t ns = 0;
	for (size_t i = 0; i < len; i++) if (seq_in[i] > 3) ns++;
	if (ns > 1) return 0;                    /* :992-998 */
	if (ns == 1 && !rep1mm) return 0;
	if (len < 2) return 0;
	/* patFw, patFwRev, patRc, patRcRev and the matching quality strings (read.h) */
	uint8_t *pat[4]; char *qu[2];
	for (int k = 0; k < 4; k++) pat[k] = (uint8_t*)malloc(len);
	for (int k = 0; k < 2; k++) qu[k] = (char*)malloc(len);
	for (size_t i = 0; i < len; i++) {
		pat[0][i] = seq_in[i];                                         /* patFw */
		pat[1][i] = seq_in[len - 1 - i];                               /* patFwRev */
		pat[2][i] = seq_in[len - 1 - i] > 3 ? 4 : 3 - seq_in[len - 1 - i];   /* patRc */
		pat[3][i] = seq_in[i] > 3 ? 4 : 3 - seq_in[i];                 /* patRcRev */
		qu[0][i] = qual_in[i]; qu[1][i] = qual_in[len - 1 - i];        /* qual, qualRev */
	}
	const size_t half_fw = len >> 1, half_bw = (len >> 1) + (len & 1);
	uint64_t t[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, tp[4] = {0, 0, 0, 0}, bp[4] = {0, 0, 0, 0};
	uint64_t top = 0, bot = 0, topp = 0, botp = 0;
	for (int fwi = 0; fwi < 2; fwi++) {
		const int fw = fwi == 0;
		if (fw && nofw) continue;
		if (!fw && norc) continue;
		const int lim = rep1mm ? 2 : 1;
		for (int ebwtfwi = 0; ebwtfwi < lim; ebwtfwi++) {
			const int ebwtfw = ebwtfwi == 0;
			const bt2o_ebwt *ebwt = ebwtfw ? ebwt_fw : ebwt_bw, *ebwtp = ebwtfw ? ebwt_bw : ebwt_fw;
			const uint8_t *seq = fw ? (ebwtfw ? pat[0] : pat[1]) : (ebwtfw ? pat[2] : pat[3]);
			const char *qual = fw ? (ebwtfw ? qu[0] : qu[1]) : (ebwtfw ? qu[1] : qu[0]);
			const int ftab_len = ebwt->ftab_chars;
			const size_t nea = ebwtfw ? half_fw : half_bw;
			int skip = 0;
			for (size_t dep = 0; dep < nea; dep++) if (seq[len - dep - 1] > 3) { skip = 1; break; }
			if (skip) continue;
			size_t dep = 0;
			if (ftab_len > 1 && (size_t)ftab_len <= nea) {            /* :1057-1072 */
				const int rev = !ebwtfw;
				bt2o_ftab_lohi(ebwt, bt2o_ftab_seq_to_int(ebwt, seq, len - ftab_len, rev), &top, &bot);
				if (rep1mm) bt2o_ftab_lohi(ebwtp, bt2o_ftab_seq_to_int(ebwtp, seq, len - ftab_len, rev), &topp, &botp);
				if (bot - top == 0) continue;
				const int c = seq[len - ftab_len];
				t[c] = top; b[c] = bot; tp[c] = topp; bp[c] = botp;
				dep = (size_t)ftab_len;
			} else {                                                  /* :1073-1084 */
				const int c = seq[len - 1];
				top = topp = tp[c] = ebwt->fchr[c];
				bot = botp = bp[c] = ebwt->fchr[c + 1];
				if (bot - top == 0) continue;
				dep = 1;
			}
			int do_continue = 0;
			for (; dep < nea; dep++) {                                /* near half: exact, :1088-1123 */
				const int rdc = seq[len - dep - 1];
				if (bot - top > 1) {
					bi_lf_ex(ebwt, top, bot, topp, t, b, tp, bp);
					top = t[rdc]; bot = b[rdc];
					if (bot <= top) { do_continue = 1; break; }
					topp = tp[rdc]; botp = bp[rdc];
				} else {
					top = bt2o_map_lf1c(ebwt, top, rdc);
					if (top == ebwt->off_mask) { do_continue = 1; break; }
					bot = top + 1;
					t[rdc] = top; b[rdc] = bot; tp[rdc] = topp; bp[rdc] = botp;
				}
			}
			if (do_continue) continue;
			for (; dep < len; dep++) {                                /* far half, :1128-1318 */
				const int rdc = seq[len - dep - 1];
				const int quc = qual[len - dep - 1];
				if (rdc > 3 && nceil == 0) break;
				tp[0] = tp[1] = tp[2] = tp[3] = topp;
				bp[0] = bp[1] = bp[2] = bp[3] = botp;
				int clo = 0, chi = 3, match = 1;
				if (bot - top > 1) {
					bi_lf_ex(ebwt, top, bot, topp, t, b, tp, bp);
					match = rdc < 4;
					if (rdc < 4) { top = t[rdc]; bot = b[rdc]; topp = tp[rdc]; botp = bp[rdc]; }   /* (rdc == 4: the reference reads past the arrays; the values are not used: match is false) */
				} else {
					clo = bt2o_map_lf1(ebwt, &top);
					match = clo == rdc;
					if (clo < 0) break;                                /* hit the $ */
					t[clo] = top; b[clo] = bot = top + 1;
					bp[clo] = botp; tp[clo] = topp;
					chi = clo;
				}
				if (rep1mm && (ns == 0 || rdc > 3)) {
					for (int j = clo; j <= chi; j++) {
						if (j == rdc || b[j] == t[j]) continue;
						size_t depm = dep + 1;
						uint64_t topm = t[j], botm = b[j], topmp = tp[j], botmp = bp[j];
						uint64_t tm[4], bm[4], tmp_[4], bmp[4];
						for (; depm < len; depm++) {
							const int rdcm = seq[len - depm - 1];
							if (botm - topm > 1) {
								bi_lf_ex(ebwt, topm, botm, topmp, tm, bm, tmp_, bmp);
								if (rdcm > 3) { topm = botm = 0; break; }       /* (cannot happen: the only N is at dep) */
								topm = tm[rdcm]; botm = bm[rdcm]; topmp = tmp_[rdcm]; botmp = bmp[rdcm];
								if (botm <= topm) break;
							} else {
								topm = bt2o_map_lf1c(ebwt, topm, rdcm);
								if (topm == ebwt->off_mask) break;
								botm = topm + 1;
							}
						}
						if (depm == len) {                             /* a 1-mismatch hit, :1213-1276 */
							size_t off5p = dep;
							if (fw == ebwtfw) off5p = len - off5p - 1;
							int64_t score = (int64_t)(len - 1) * match_bonus;
							const int pen = bt2o_score(sc, rdc, 1 << j, quc - 33);
							score += pen;
							int valid = 1;
							if (local) {
								int64_t lf = 0, lb = 0;
								for (size_t i = 0; i < len; i++) {
									if (i == dep) { if (lf + pen <= 0) { valid = 0; break; } lf += pen; } else lf += match_bonus;
									if (len - i - 1 == dep) { if (lb + pen <= 0) { valid = 0; break; } lb += pen; } else lb += match_bonus;
								}
							}
							if (valid) valid = score >= minsc;
							if (valid) {
								if (nout < cap) {
									bt2o_mm1_hit *h = &out[nout];
									memset(h, 0, sizeof(*h));
									h->top = ebwtfw ? topm : topmp; h->bot = ebwtfw ? botm : botmp;
									h->score = score; h->off5p = (uint32_t)off5p; h->chr = (uint8_t)j; h->qchr = (uint8_t)rdc;
									h->fw = (uint8_t)fw; h->kind = 1; h->ebwtfw = (uint8_t)ebwtfw;
								}
								nout++;
							}
						}
					}
				}
				if (bot > top && match) {
					if (dep == len - 1) {
						if (ebwtfw && repex) {                         /* an exact hit, :1285-1305 */
							if (nout < cap) {
								bt2o_mm1_hit *h = &out[nout];
								memset(h, 0, sizeof(*h));
								h->top = top; h->bot = bot; h->score = (int64_t)len * match_bonus; h->fw = (uint8_t)fw; h->kind = 0; h->ebwtfw = 1;
							}
							nout++;
						}
						break;
					}
				} else break;
			}
		}
	}
	for (int k = 0; k < 4; k++) free(pat[k]);
	for (int k = 0; k < 2; k++) free(qu[k]);
	return nout;
}

/* ------------------------------------------------------------------ */
/* reference fetch                                                     */
/* ------------------------------------------------------------------ */
int bt2o_ref_get_base(const bt2o_ref *r, uint64_t tidx, uint64_t toff) {
	uint64_t reci = r->ref_rec_offs[tidx], recf = r->ref_rec_offs[tidx + 1];
	uint64_t buf_off = r->ref_offs[tidx], off = 0;
	for (uint64_t i = reci; i < recf; i++) {
		off += r->rec_off[i];
		if (toff < off) return 4;
		uint64_t rec_end = off + r->rec_len[i];
		if (toff < rec_end) {
			uint64_t bo = buf_off + (toff - off);
			return (r->buf[bo >> 2] >> ((bo & 3) << 1)) & 3;
		}
		buf_off += r->rec_len[i];
		off = rec_end;
	}
	return 4;
}

void bt2o_ref_get_stretch(const bt2o_ref *r, uint8_t *dest, uint64_t tidx, int64_t toff, size_t count) {
	for (size_t i = 0; i < count; i++) {
		int64_t p = toff + (int64_t)i;
		dest[i] = (p < 0 || (uint64_t)p >= r->ref_lens[tidx]) ? 4 : (uint8_t)bt2o_ref_get_base(r, tidx, (uint64_t)p);
	}
}

/* ------------------------------------------------------------------ */
/* RNG                                                                 */
/* ------------------------------------------------------------------ */
void bt2o_rng_init(bt2o_rng *r, uint32_t seed) { r->a = 1664525u; r->c = 1013904223u; r->last = seed; r->lastOff = 30; r->inited = 1; }
uint32_t bt2o_rng_next_u32(bt2o_rng *r) {
	r->last = r->a * r->last + r->c;
	uint32_t ret = r->last >> 16;
	r->last = r->a * r->last + r->c;
	ret ^= r->last;
	r->lastOff = 0;
	return ret;
}
uint64_t bt2o_rng_next_u64(bt2o_rng *r) {
	uint64_t hi = bt2o_rng_next_u32(r);
	uint64_t lo = bt2o_rng_next_u32(r);
	return (hi << 32) | lo;
}
uint32_t bt2o_rng_next_u2(bt2o_rng *r) {
	if (r->lastOff > 30) bt2o_rng_next_u32(r);
	uint32_t ret = (r->last >> r->lastOff) & 3;
	r->lastOff += 2;
	return ret;
}
int bt2o_rng_next_bool(bt2o_rng *r) {
	if (r->lastOff > 31) bt2o_rng_next_u32(r);
	uint32_t ret = (r->last >> r->lastOff) & 1;
	r->lastOff++;
	return (int)ret;
}
float bt2o_rng_next_float(bt2o_rng *r) { return (float)bt2o_rng_next_u32(r) / (float)0xffffffffu; }

uint32_t bt2o_gen_rand_seed(const uint8_t *seq, const char *qual, size_t len,
                            const char *name, size_t namelen, uint32_t seed) {
	uint32_t rseed = (seed + 101u) * 59u * 61u * 67u * 71u * 73u * 79u * 83u;
	for (size_t i = 0; i < len; i++) rseed ^= ((uint32_t)seq[i] << ((i & 15) << 1));
	for (size_t i = 0; i < len; i++) rseed ^= ((uint32_t)(int)qual[i] << ((i & 3) << 3));
	for (size_t i = 0; i < namelen; i++) {
		int p = (int)name[i];
		if (p == '/') break;
		rseed ^= ((uint32_t)p << ((i & 3) << 3));
	}
	return rseed;
}

/* ------------------------------------------------------------------ */
/* scoring + end-to-end u8 DP fill                                     */
/* ------------------------------------------------------------------ */
void bt2o_scoring_default(bt2o_scoring *sc) {
	sc->match_bonus = 0; sc->mm_pen_type = 3; sc->mm_max = 6; sc->mm_min = 2; sc->n_pen = 1;
	sc->rd_gap_const = 5; sc->rd_gap_linear = 3; sc->rf_gap_const = 5; sc->rf_gap_linear = 3;
	sc->gapbar = 4;
}

static int mm_pen(const bt2o_scoring *sc, int q) {
	if (sc->mm_pen_type == 3) { /* COST_MODEL_QUAL, scoring.h:106-114 */
		int ii = q < 40 ? q : 40;
		float frac = (float)ii / 40.0f;
		return sc->mm_min + (int)(frac * (float)(sc->mm_max - sc->mm_min));
	}
	return sc->mm_max;
}

int bt2o_score(const bt2o_scoring *sc, int rdc, int refmask, int q) {
	if (q < 0) q = 0;
	if (q > 255) q = 255;
	if (rdc > 3 || refmask > 15) return -sc->n_pen;
	if (refmask & (1 << rdc)) return sc->match_bonus;
	return -mm_pen(sc, q);
}

static inline int subs(int a, int b) { int r = a - b; return r < 0 ? 0 : r; }
static inline int max2(int a, int b) { return a > b ? a : b; }

int bt2o_sw_fill_ee_u8(const bt2o_scoring *sc, const uint8_t *rd, const uint8_t *qu, int rows,
                       const uint8_t *rf, int cols, uint8_t *H, uint8_t *E, uint8_t *F) {
	const int rdgapo = sc->rd_gap_const + sc->rd_gap_linear, rdgape = sc->rd_gap_linear;
	const int rfgapo = sc->rf_gap_const + sc->rf_gap_linear, rfgape = sc->rf_gap_linear;
	int lrmax = 0;
	for (int j = 0; j < cols; j++) {
		/* the profile row is picked by the lowest set bit of the mask, N (16) -> row 4
		 * (aligner_swsse_ee_u8.cpp:917-919, mask.cpp:31) */
		int m = rf[j], refc = 4;
		for (int b = 0; b < 5; b++) if (m & (1 << b)) { refc = b; break; }
		int f = 0;
		for (int i = 0; i < rows; i++) {
			int veto = (i < sc->gapbar || rows - i - 1 < sc->gapbar) ? 0xff : 0;
			int pen = -bt2o_score(sc, rd[i], 1 << refc, qu[i]);
			int hdiag = (i == 0) ? 0xff : (j == 0 ? 0 : H[(i - 1) * cols + (j - 1)]);
			int e = (j == 0) ? 0
			      : max2(subs(E[i * cols + j - 1], rdgape),
			             subs(subs(H[i * cols + j - 1], rdgapo), veto));
			/* F[i] = max(F[i-1]-ext, H[i-1]-open) -sat veto_i ; F[0] = 0 */
			f = (i == 0) ? 0 : subs(max2(subs(f, rfgape), subs(H[(i - 1) * cols + j], rfgapo)), veto);
			int h = max2(max2(subs(hdiag, pen), e), f);
			H[i * cols + j] = (uint8_t)h; E[i * cols + j] = (uint8_t)e; F[i * cols + j] = (uint8_t)f;
		}
		if (H[(rows - 1) * cols + j] > lrmax) lrmax = H[(rows - 1) * cols + j];
	}
	return lrmax - 0xff;
}

/* ------------------------------------------------------------------ */
/* the other three DP fills of the path                                */
/* ------------------------------------------------------------------ */
/* kind 1: end-to-end 16-bit  (alignNucleotidesEnd2EndSseI16, aligner_swsse_ee_i16.cpp:780-1170)
 * kind 2: local 8-bit        (alignNucleotidesLocalSseU8,    aligner_swsse_loc_u8.cpp:927-1330)
 * kind 3: local 16-bit       (alignNucleotidesLocalSseI16,   aligner_swsse_loc_i16.cpp:938-1375)
 * The striped SSE kernels compute, cell by cell, the recurrences below in SATURATING arithmetic over the word's range [lo, hi]:
 * unsigned 8-bit for kind 2 (a biased profile: +(sc+bias) then -bias, so an overflow clips at 255-bias), signed 16-bit with
 * 0x8000 standing for "zero"/"minus infinity" for kinds 1 and 3.  A gap-barrier row forces the gap term to lo (the 0xff subtrahend,
 * or 0x8000 added twice).  Cells are stored as the kernels hold them (SSEMatrix::elt), so they can be compared word for word.
 * Local fills stop early: at a column whose maximum saturates (flag -2), or when the rest of the columns cannot lift the score
 * to minsc any more (colstop); cells of later columns are not written.
 * Returns the kernel's return value; *flag: 0 ok, -1 below minsc, -2 saturated; *colstop: columns filled. */
static inline int sat_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int64_t bt2o_sw_fill_kind(int kind, const bt2o_scoring *sc, const uint8_t *rd, const uint8_t *qu, int rows,
                          const uint8_t *rf, int cols, int64_t minsc, int32_t *H, int32_t *E, int32_t *F, int *flag, int *colstop) {
	const int rdgapo = sc->rd_gap_const + sc->rd_gap_linear, rdgape = sc->rd_gap_linear;
	const int rfgapo = sc->rf_gap_const + sc->rf_gap_linear, rfgape = sc->rf_gap_linear;
	const int lo = kind == 2 ? 0 : -32768, hi = kind == 2 ? 255 : 32767;
	const int local = kind != 1;
	int bias = 0;
	if (kind == 2) {     /* buildQueryProfileLocalSseU8: the largest penalty any (read position, reference character) pair can draw */
		for (int refc = 0; refc < 5; refc++)
			for (int i = 0; i < rows; i++) { int s = bt2o_score(sc, rd[i], 1 << refc, qu[i]); if (-s > bias) bias = -s; }
	}
	int64_t vmax = lo, lrmax = lo;
	*flag = 0;
	*colstop = cols;
	for (int j = 0; j < cols; j++) {
		int m = rf[j], refc = 4;
		for (int b = 0; b < 5; b++) if (m & (1 << b)) { refc = b; break; }       /* firsts5[mask] */
		int f = lo, colmax = lo;
		for (int i = 0; i < rows; i++) {
			const int barrier = (i < sc->gapbar || rows - i - 1 < sc->gapbar);
			const int s = bt2o_score(sc, rd[i], 1 << refc, qu[i]);
			int hdiag = (i == 0) ? (local ? lo : hi) : (j == 0 ? lo : H[(i - 1) * cols + (j - 1)]);
			int hd = kind == 2 ? sat_i(sat_i(hdiag + s + bias, lo, hi) - bias, lo, hi) : sat_i(hdiag + s, lo, hi);
			int e = lo;
			if (j > 0) {
				int open = barrier ? lo : sat_i(H[i * cols + j - 1] - rdgapo, lo, hi);
				int ext = sat_i(E[i * cols + j - 1] - rdgape, lo, hi);
				e = open > ext ? open : ext;
			}
			if (i == 0) f = lo;
			else {
				int open = sat_i(H[(i - 1) * cols + j] - rfgapo, lo, hi), ext = sat_i(f - rfgape, lo, hi);
				f = barrier ? lo : (open > ext ? open : ext);
			}
			int h = hd;
			if (e > h) h = e;
			if (f > h) h = f;
			H[i * cols + j] = h; E[i * cols + j] = e; F[i * cols + j] = f;
			if (h > colmax) colmax = h;
		}
		if (H[(rows - 1) * cols + j] > lrmax) lrmax = H[(rows - 1) * cols + j];
		if (colmax > vmax) vmax = colmax;
		if (local) {
			if (kind == 2 && colmax + bias >= 255) { *flag = -2; *colstop = j + 1; return INT64_MIN; }
			const int64_t score = kind == 2 ? colmax : (int64_t)colmax + 32768;
			if (score < minsc && score + (int64_t)(cols - j - 1) * sc->match_bonus < minsc) { *colstop = j + 1; break; }
		}
	}
	if (kind == 1) {
		const int64_t score = lrmax - 0x7fff;
		if (score < minsc) { *flag = -1; return score; }
		if (lrmax == -32768) { *flag = -2; return INT64_MIN; }
		return score;
	}
	if (kind == 2) {
		if (vmax + bias >= 255) { *flag = -2; return INT64_MIN; }
		if (vmax == 0 || vmax < minsc) { *flag = -1; return vmax; }
		return vmax;
	}
	if (vmax == -32768) { *flag = -1; return INT64_MIN; }
	if (vmax + 32768 < minsc) { *flag = -1; return vmax + 32768; }
	if (vmax == 32767) { *flag = -2; return INT64_MIN; }
	return vmax + 32768;
}
