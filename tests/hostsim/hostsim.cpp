// tests/hostsim/hostsim.cpp -- TEST-ONLY host build of the per-read worker.
//
// Compiles bowtie2_amd/csrc/bt2g_align_core.hpp for the CPU (the wave-parallel sections
// replaced by plain loops writing the same wavefront-major scratch layout) so that the
// control logic can be diffed against the reference's SAM in a container without a GPU.
// It is never linked into libbt2g.so or the bowtie2-align-* drop-in: the product path is
// the HIP kernel only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../bowtie2_amd/csrc/bt2g_index.hpp"
#include "../../bowtie2_amd/csrc/bt2g_align_core.hpp"
#include "../../bowtie2_amd/csrc/bt2g_local_pk.hpp"
#include "../../bowtie2_amd/csrc/bt2g_rankidx.hpp"
#include "../../bowtie2_amd/csrc/bt2g_host.hpp"
#include "../../bowtie2_amd/csrc/bt2g_pipeline.hpp"
#include "../../bowtie2_amd/csrc/bt2g_cli.hpp"
#include <type_traits>
static_assert(std::is_trivially_default_constructible<bt2g::Work>::value && std::is_trivially_destructible<bt2g::Work>::value, "Work is allocated with calloc");

using namespace bt2g;

static HotWork g_hot;
static uint8_t host_rf[kMaxColsWide + 8];          // the per-column tail of the hot state (dynamic LDS on the device)
static int16_t host_lastrow[kMaxColsWide + 8];
static Edit host_ned[kMaxWalkEdits];
static AlState g_st;
static const AlignParams* g_Pp = nullptr;      // the control blocks the device keeps in LDS
static ReadParams g_rp;
static const void* g_ixp = nullptr;
static CliExtra g_ex;

struct HostPlat {
	static HotWork& hot() { return g_hot; }
	static uint8_t* rf() { return host_rf; }
	static Edit* ned() { return host_ned; }
	static int16_t* lastrow() { return host_lastrow; }
	static const AlignParams& params() { return *g_Pp; }
	static ReadParams& rparams() { return g_rp; }
	static const PreComp* pre() { return nullptr; }
	static AlState& st() { return g_st; }
	static Work& work() { return *g_st.wp; }
	template <typename TOff> static const DevIndex<TOff>& index() { return *reinterpret_cast<const DevIndex<TOff>*>(g_ixp); }
	static uint64_t clock() { return 0; }
	template <typename T> static T uni(T v) { return v; }
	// lane-strided loops of the worker (one "lane" on the host)
	static uint32_t lane_id() { return 0; }
	static uint32_t n_lanes() { return 1; }
	static bool any(bool b) { return b; }
	static void sync() {}
	static std::vector<uint16_t>& local_h() { static std::vector<uint16_t> v; return v; }      // H of the last local fill (host only)
	static std::vector<uint8_t>& local_ef() { static std::vector<uint8_t> v; return v; }        // BT2G_CHECK_LOCAL_PK: bit 0 = E > 0, bit 1 = F > 0 of the last local fill
	template <typename T> static T* uni_ptr(T* p) { return p; }
	template <typename TOff> static TOff get_offset(const DevEbwt<TOff>& e, TOff row, uint32_t& nsteps) { return bt2g::get_offset(e, row, nsteps); }
	static void fetch_ref(const DevRef& ref, Work& w, uint64_t tidx, int64_t rfi, uint32_t count) {
		const uint64_t rec0 = ref_rec_find(ref, tidx, rfi);
		for (uint32_t i = 0; i < count; i++) host_rf[i] = (uint8_t)(1 << ref_base_at(ref, tidx, rfi + (int64_t)i, rec0));
	}
	static void fetch_ref_codes(const DevRef& ref, uint64_t tidx, int64_t rfi, uint32_t count) {
		const uint64_t rec0 = ref_rec_find(ref, tidx, rfi);
		for (uint32_t i = 0; i < count; i++) host_rf[i] = (uint8_t)ref_base_at(ref, tidx, rfi + (int64_t)i, rec0);
	}
	static void zero_masks(uint16_t* p, uint32_t n) { memset(p, 0, (size_t)n * 2); }
	static void zero_u32(uint32_t* p, uint32_t n) { memset(p, 0, (size_t)n * 4); }
	static void set_epoch(uint32_t* p, uint32_t e) { *p = e; }
	template <typename TOff> static void resolve_rows(const DevEbwt<TOff>& e, const SampRow* rows, uint32_t n, uint64_t* out) {
		for (uint32_t l = 0; l < n; l++) { uint32_t steps = 0; const TOff joff = bt2g::get_offset(e, (TOff)rows[l].topf, steps); out[l] = joff_pack((uint64_t)joff, steps); }
	}
	static bool contains_u32(const uint32_t* p, uint32_t n, uint32_t v) { for (uint32_t i = 0; i < n; i++) if (p[i] == v) return true; return false; }
	static void unseen_list(const uint32_t* seen, uint32_t nseen, uint32_t n, uint32_t* out) {
		std::vector<bool> in(n, false);
		for (uint32_t i = 0; i < nseen; i++) if (seen[i] < n) in[seen[i]] = true;
		uint32_t c = 0;
		for (uint32_t j = 0; j < n; j++) if (!in[j]) out[c++] = j;
	}
	static void iota_u32(uint32_t* p, uint32_t n) { for (uint32_t i = 0; i < n; i++) p[i] = i; }
	static void order_by_score(const AlnRes* alns, uint32_t n, uint32_t* idx, uint32_t*) {
		for (uint32_t i = 0; i < n; i++) idx[i] = i;
		std::sort(idx, idx + n, [&](uint32_t a, uint32_t b) { return alns[a].score != alns[b].score ? alns[a].score > alns[b].score : a > b; });      // descending by (score, index)
	}
	static void copy_words(void* dst, const void* src, uint32_t nwords) { memcpy(dst, src, (size_t)nwords * 4); }
	static void load_last_row(const uint32_t* mat, uint32_t R, uint32_t rows, uint32_t cols, bool wide) {
		for (uint32_t j = 0; j < cols; j++) {
			int sc;
			if (wide) sc = (int)(int16_t)(uint16_t)(reinterpret_cast<const uint64_t*>(mat)[dp_cell(R, rows - 1, j)] & 0xffff) - 0x7fff;
			else sc = (int)(mat[dp_cell(R, rows - 1, j)] & 0xff) - 0xff;
			host_lastrow[j] = (int16_t)(sc < -32768 ? -32768 : sc);
		}
	}
	static void load_read(const uint8_t* seq, const uint8_t* qual, uint32_t len) { memcpy(g_hot.seq, seq, len); memcpy(g_hot.qual, qual, len); }
	static void copy_aln(AlnRes& dst, const AlnRes& src) { memcpy(&dst, &src, offsetof(AlnRes, ned) + (size_t)src.nned * sizeof(Edit)); }
	static uint32_t gather_sort(BtCand* cands, uint32_t cap, uint32_t rows, uint32_t cols, int64_t minsc_dp) {
		uint32_t n = 0, total = 0;
		for (uint32_t j = 0; j < cols; j++) {
			const int sc = (int)host_lastrow[j];
			if (sc < minsc_dp) continue;
			total++;
			if (n >= cap) continue;
			BtCand c; c.score = sc; c.row = (uint16_t)(rows - 1); c.col = (uint16_t)j;
			uint32_t k = n++;      // insertion keeps: score desc, col desc
			while (k > 0 && (cands[k - 1].score < c.score || (cands[k - 1].score == c.score && cands[k - 1].col < c.col))) { cands[k] = cands[k - 1]; k--; }
			cands[k] = c;
		}
		return total;
	}
	struct LaneReg { uint32_t v[64]; };
	static uint32_t lane(const LaneReg& r, uint32_t i) { return r.v[i]; }
	static void set_lane(LaneReg& r, uint32_t i, uint32_t v) { r.v[i] = v; }
	static void lanes_zero(LaneReg& r) { memset(r.v, 0, sizeof(r.v)); }
	static LaneReg lanes_load(const uint8_t* base, uint32_t nbytes, uint32_t word0) {
		LaneReg r;
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t wd = word0 + l;
			uint32_t v = 0;
			if (wd * 4 + 4 <= nbytes) memcpy(&v, base + wd * 4, 4);
			r.v[l] = v;
		}
		return r;
	}
	// ---- the register-only row sampler (Aligner::sample_rows_fast): plain-loop forms of the lane-parallel primitives ----
	static void samp_setup(const SatPos* sat, uint32_t n, bool all_hits, LaneReg& mlo, LaneReg& mhi, LaneReg& rn, LaneReg& rthr, LaneReg& rfl, LaneReg& tlo, LaneReg& thi) {
		for (uint32_t l = 0; l < 64; l++) {
			double m = 0.0;
			rn.v[l] = rthr.v[l] = rfl.v[l] = tlo.v[l] = thi.v[l] = 0;
			if (l < n) {
				m = samp_mass(sat[l].nlex, sat[l].nrex, sat[l].size);
				const uint32_t sz = sat[l].size;
				uint32_t th = (uint32_t)(0.10f * (float)sz); th = th > 16 ? th : 16;      // Random1toN::init (random_util.h:97-110)
				rn.v[l] = sz; rthr.v[l] = th; rfl.v[l] = (sz < 128 || all_hits) ? 1u : 0u;
				tlo.v[l] = (uint32_t)sat[l].topf; thi.v[l] = (uint32_t)(sat[l].topf >> 32);
			}
			uint64_t u; memcpy(&u, &m, 8);
			mlo.v[l] = (uint32_t)u; mhi.v[l] = (uint32_t)(u >> 32);
		}
	}
	static double prefix_live(const LaneReg& mlo, const LaneReg& mhi, uint64_t live, LaneReg& plo, LaneReg& phi) {
		double acc = 0.0;
		for (uint32_t i = 0; i < 64; i++) if ((live >> i) & 1ull) { acc += f64_of(mlo.v[i], mhi.v[i]); uint64_t u; memcpy(&u, &acc, 8); plo.v[i] = (uint32_t)u; phi.v[i] = (uint32_t)(u >> 32); }
		return acc;
	}
	static uint32_t pick_prefix(const LaneReg& plo, const LaneReg& phi, uint64_t live, double rd) {
		uint32_t last = 0xffffffffu;
		for (uint32_t i = 0; i < 64; i++) if ((live >> i) & 1ull) { last = i; if (rd < f64_of(plo.v[i], phi.v[i])) return i; }
		return last;
	}
	// (the host keeps entry e in lane e & 63 of register e >> 6)
	template <int K> struct LaneRegs { LaneReg r[K]; LaneReg& operator[](uint32_t i) { return r[i]; } const LaneReg& operator[](uint32_t i) const { return r[i]; } };
	template <int K> static void tab_zero(LaneRegs<K>& t) { memset(&t, 0, sizeof(t)); }
	template <typename T> static bool tab_lookup(const T& k, const T& v, uint32_t n, uint32_t key, uint32_t& val) {
		for (uint32_t i = 0; i < n; i++) if (k[i >> 6].v[i & 63] == key) { val = v[i >> 6].v[i & 63]; return true; }
		return false;
	}
	template <typename T> static void tab_set(const T& k, T& v, uint32_t n, uint32_t key, uint32_t val) {
		for (uint32_t i = 0; i < n; i++) if (k[i >> 6].v[i & 63] == key) { v[i >> 6].v[i & 63] = val; return; }
	}
	template <typename T> static void tab_append(T& k, T& v, uint32_t& n, uint32_t key, uint32_t val) { k[n >> 6].v[n & 63] = key; v[n >> 6].v[n & 63] = val; n++; }
	template <typename T> static uint32_t tab_count_le(const T& k, const T& v, uint32_t n, uint32_t keyhi, uint32_t x) {
		uint32_t c = 0;
		for (uint32_t i = 0; i < n; i++) if ((k[i >> 6].v[i & 63] & 0xff000000u) == keyhi && v[i >> 6].v[i & 63] <= x) c++;
		return c;
	}
	// the seen values of `range` (kind 2 entries) become kind 3 entries carrying value - rank
	template <typename T> static void tab_convert(T& k, T& v, uint32_t n, uint32_t range) {
		const uint32_t seenhi = (2u << 30) | (range << 24), convhi = (3u << 30) | (range << 24);
		for (uint32_t i = 0; i < n; i++) {
			const uint32_t ki = k[i >> 6].v[i & 63];
			if ((ki & 0xff000000u) != seenhi) continue;
			uint32_t rank = 0;
			for (uint32_t j = 0; j < n; j++) { const uint32_t kj = k[j >> 6].v[j & 63]; if ((kj & 0xff000000u) == seenhi && (kj & 0xffffffu) < (ki & 0xffffffu)) rank++; }
			v[i >> 6].v[i & 63] = (ki & 0xffffffu) - rank;
		}
		for (uint32_t i = 0; i < n; i++) { uint32_t& ki = k[i >> 6].v[i & 63]; if ((ki & 0xff000000u) == seenhi) ki = convhi | (ki & 0xffffffu); }
	}
	// lane j <- the fields of range sat[j] the extension loop needs of a sampled row: rdoff | seedlen << 12 | fw << 18 | offidx << 20
	static LaneReg range_fields(const SatPos* sat, uint32_t n) {
		LaneReg r;
		for (uint32_t l = 0; l < 64; l++) r.v[l] = l < n ? (sat[l].rdoff & 0xfffu) | ((sat[l].seedlen & 0x3fu) << 12) | ((uint32_t)(sat[l].fw != 0) << 18) | (sat[l].offidx << 20) : 0u;
		return r;
	}
	// ---- lane code (BT2_FOR_LANES / LV in bt2g_align_core.hpp): on the host a loop over the 64 lanes of LaneReg arrays ----
	static uint32_t lanes_first() { return 0; }
	static uint32_t lanes_step() { return 1; }
	static uint32_t& lv(LaneReg& r, uint32_t l) { return r.v[l]; }
	static const uint32_t& lv(const LaneReg& r, uint32_t l) { return r.v[l]; }
	static bool ee_wide_band(uint32_t, uint32_t, int64_t) { return false; }      // (the twin fills 16-bit end-to-end problems in the cell format)
	static uint64_t lanes_sum(const LaneReg& r) { uint64_t v = 0; for (uint32_t l = 0; l < 64; l++) v += r.v[l]; return v; }
	static uint64_t ballot(const LaneReg& r) { uint64_t m = 0; for (uint32_t l = 0; l < 64; l++) if (r.v[l]) m |= 1ull << l; return m; }
	static LaneReg gather(const LaneReg& x, const LaneReg& idx) { LaneReg r; for (uint32_t l = 0; l < 64; l++) r.v[l] = x.v[idx.v[l] & 63u]; return r; }
	template <typename T, typename F> static void tab_for_each(const T& k, const T& v, uint32_t n, F f, uint32_t from = 0) { for (uint32_t e = from; e < n; e++) f(e, k[e >> 6].v[e & 63], v[e >> 6].v[e & 63]); }
	template <typename T> static void tab_set_at(T& v, uint32_t e, uint32_t val) { v[e >> 6].v[e & 63] = val; }
	// the flagged lanes' (key, value) pairs become the next entries of the table, in lane order
	template <typename T> static void tab_append_lanes(T& k, T& v, uint32_t& n, const LaneReg& flag, const LaneReg& key, const LaneReg& val) {
		for (uint32_t l = 0; l < 64; l++) if (flag.v[l]) { k[n >> 6].v[n & 63] = key.v[l]; v[n >> 6].v[n & 63] = val.v[l]; n++; }
	}
	// the sampler's hash tables (dynamic LDS on the device): the same open addressing with the same hash on host arrays.  BT2G_HOST_SH_BYTES = the dynamic
	// LDS a launch would have (default: the short-read class's headline launch)
	static constexpr uint32_t kShBatchSlots = 96u;
	struct WSlot { uint32_t k, v; };
	struct BhSlot { uint32_t k; uint64_t m; };
	static uint32_t* sh_S() { static uint32_t t[4096]; return t; }
	static WSlot* sh_W() { static WSlot t[1024]; return t; }
	static BhSlot* sh_B() { static BhSlot t[kShBatchSlots]; return t; }
	static uint32_t sh_slot0(uint32_t key, uint32_t nslots) { return (((key * 2654435761u) >> 16) * nslots) >> 16; }
	static uint32_t sh_begin() {
		static const uint32_t have = getenv("BT2G_HOST_SH_BYTES") ? (uint32_t)atoi(getenv("BT2G_HOST_SH_BYTES")) : 4528u;
		if (have < kShBatchSlots * 12u + 1024u) return 0u;
		const uint32_t room = have - kShBatchSlots * 12u;
		uint32_t nW = (room * 3u / 5u) / 8u, nS = (room - nW * 8u) / 4u;
		if (nW > 1024u) nW = 1024u;
		if (nS > 4096u) nS = 4096u;
		memset(sh_S(), 0, nS * 4u); memset(sh_W(), 0, sizeof(WSlot) * nW);
		return nS | (nW << 16);
	}
	static void bh_clear(uint32_t) { for (uint32_t i = 0; i < kShBatchSlots; i++) { sh_B()[i].k = 0; sh_B()[i].m = 0; } }
	static BhSlot* bh_find(uint32_t key, bool claim) {
		uint32_t s = sh_slot0(key, kShBatchSlots);
		for (uint32_t pr = 0; pr < kShBatchSlots; pr++) {
			if (sh_B()[s].k == key) return &sh_B()[s];
			if (sh_B()[s].k == 0u) { if (!claim) return nullptr; sh_B()[s].k = key; return &sh_B()[s]; }
			s = s + 1u == kShBatchSlots ? 0u : s + 1u;
		}
		return nullptr;
	}
	static void sh_put1(uint32_t cfg, uint32_t key, uint32_t idx) {
		const uint32_t nS = cfg & 0xffffu, nW = cfg >> 16;
		if ((key >> 30) == 2u) {
			uint32_t s = sh_slot0(key, nS);
			for (uint32_t pr = 0; pr < nS; pr++) { if (sh_S()[s] == key) return; if (sh_S()[s] == 0u) { sh_S()[s] = key; return; } s = s + 1u == nS ? 0u : s + 1u; }
		} else {
			uint32_t s = sh_slot0(key, nW);
			for (uint32_t pr = 0; pr < nW; pr++) { if (sh_W()[s].k == key || sh_W()[s].k == 0u) { sh_W()[s].k = key; sh_W()[s].v = idx; return; } s = s + 1u == nW ? 0u : s + 1u; }
		}
	}
	static void sh_put(uint32_t cfg, const LaneReg& key, const LaneReg& idx, const LaneReg& flag) { for (uint32_t l = 0; l < 64; l++) if (flag.v[l]) sh_put1(cfg, key.v[l], idx.v[l]); }
	static void sh_get(uint32_t cfg, const LaneReg& key, const LaneReg& flag, LaneReg& found, LaneReg& idx) {
		const uint32_t nS = cfg & 0xffffu, nW = cfg >> 16;
		for (uint32_t l = 0; l < 64; l++) {
			found.v[l] = idx.v[l] = 0;
			if (!flag.v[l]) continue;
			const uint32_t k = key.v[l];
			if ((k >> 30) == 2u) {
				uint32_t s = sh_slot0(k, nS);
				for (uint32_t pr = 0; pr < nS; pr++) { if (sh_S()[s] == k) { found.v[l] = 1; break; } if (sh_S()[s] == 0u) break; s = s + 1u == nS ? 0u : s + 1u; }
			} else {
				uint32_t s = sh_slot0(k, nW);
				for (uint32_t pr = 0; pr < nW; pr++) { if (sh_W()[s].k == k) { found.v[l] = 1; idx.v[l] = sh_W()[s].v; break; } if (sh_W()[s].k == 0u) break; s = s + 1u == nW ? 0u : s + 1u; }
			}
		}
	}
	static void bh_mark(uint32_t, const LaneReg& key, const LaneReg& flag) {
		for (uint32_t l = 0; l < 64; l++) if (flag.v[l]) { BhSlot* p = bh_find(key.v[l], true); if (p) p->m |= 1ull << l; }
	}
	static void bh_get(uint32_t, const LaneReg& key, const LaneReg& flag, LaneReg& lo, LaneReg& hi) {
		for (uint32_t l = 0; l < 64; l++) { lo.v[l] = hi.v[l] = 0; if (flag.v[l]) { BhSlot* p = bh_find(key.v[l], false); if (p) { lo.v[l] = (uint32_t)p->m; hi.v[l] = (uint32_t)(p->m >> 32); } } }
	}
	template <typename T> static LaneReg tab_gather(const T& v, const LaneReg& idx) { LaneReg r; for (uint32_t l = 0; l < 64; l++) r.v[l] = v[(idx.v[l] >> 6) & 7u].v[idx.v[l] & 63]; return r; }
	static void flush_samp_rows(SampRow* dst, const LaneReg& lo, const LaneReg& hi, const LaneReg& src, uint32_t cnt) {
		for (uint32_t l = 0; l < cnt; l++) { dst[l].topf = ((uint64_t)hi.v[l] << 32) | lo.v[l]; dst[l].src = src.v[l]; dst[l].done = 0; }
	}
	static LaneReg lanes_load_u32(const uint32_t* p, uint32_t base, uint32_t n) {
		LaneReg r;
		for (uint32_t l = 0; l < 64; l++) r.v[l] = base + l < n ? p[base + l] : 0u;
		return r;
	}
	static void lanes_load_keys(const uint64_t* keys, const uint8_t* lens, const uint8_t* flags, const uint32_t* eff, uint32_t n, LaneReg& klo, LaneReg& khi, LaneReg& klf, LaneReg& kef) {
		for (uint32_t l = 0; l < 64; l++) {
			klo.v[l] = khi.v[l] = klf.v[l] = kef.v[l] = 0;
			if (l < n) { klo.v[l] = (uint32_t)keys[l]; khi.v[l] = (uint32_t)(keys[l] >> 32); klf.v[l] = (uint32_t)lens[l] | ((uint32_t)flags[l] << 8); kef.v[l] = eff[l]; }
		}
	}
	static uint32_t find_key_lanes(const LaneReg& klo, const LaneReg& khi, const LaneReg& klf, uint32_t n, uint64_t key, uint8_t len) {
		for (uint32_t l = 0; l < n && l < 64; l++) if (klo.v[l] == (uint32_t)key && khi.v[l] == (uint32_t)(key >> 32) && (klf.v[l] & 0xffu) == (uint32_t)len) return l;
		return n;
	}
	// lane i holds candidate i of a batch as row | col << 16: flag the ones within sq rows and columns of the cell rc
	static void dom_update(LaneReg& domv, const LaneReg& cw1, uint32_t rc, uint32_t sq) {
		const uint32_t orow = rc & 0xffffu, ocol = rc >> 16;
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t row = cw1.v[l] & 0xffffu, col = cw1.v[l] >> 16;
			const uint32_t dr = row > orow ? row - orow : orow - row, dc = col > ocol ? col - ocol : ocol - col;
			if (dr <= sq && dc <= sq) domv.v[l] = 1;
		}
	}
	// first candidate (lane) >= from of the nv in the batch that is not flagged, nv if none; low_first: one whose score is below minsc comes first
	static uint32_t next_cand(const LaneReg& cw0, const LaneReg& domv, uint32_t from, uint32_t nv, int64_t minsc, bool& low_first) {
		low_first = false;
		for (uint32_t l = from; l < nv; l++) {
			if ((int64_t)(int32_t)cw0.v[l] < minsc) { low_first = true; return l; }
			if (!domv.v[l]) return l;
		}
		return nv;
	}
	static bool near_any(const LaneReg& r, uint32_t n, uint32_t row, uint32_t col, uint32_t sq) {
		for (uint32_t l = 0; l < n && l < 64; l++) {
			const uint32_t orow = r.v[l] & 0xffffu, ocol = r.v[l] >> 16;
			const uint32_t dr = row > orow ? row - orow : orow - row, dc = col > ocol ? col - ocol : ocol - col;
			if (dr <= sq && dc <= sq) return true;
		}
		return false;
	}
	static void lanes_load_cands(const BtCand* cands, uint32_t base, uint32_t n, LaneReg& w0, LaneReg& w1) {
		for (uint32_t l = 0; l < 64; l++) {
			const uint32_t i = base + l;
			w0.v[l] = i < n ? (uint32_t)cands[i].score : 0u;
			w1.v[l] = i < n ? ((uint32_t)cands[i].row | ((uint32_t)cands[i].col << 16)) : 0u;
		}
	}
	static void bt_tile(const DpScratch& dp, uint32_t R, uint32_t cols, uint32_t row, uint32_t col, bool wide, LaneReg& lo, LaneReg& hi) {
		for (uint32_t ln = 0; ln < 64; ln++) {
			const uint32_t d = ln & 15, g = ln >> 4;
			const int r = (int)row - (int)d - (g == 1 ? 1 : 0);
			const int c = (int)col - (int)d - (g == 2 ? 1 : 0);
			uint32_t v = 0, vh = 0;
			if (r >= 0 && c >= 0) {
				if (g < 3) {
					if (wide) { const uint64_t x = reinterpret_cast<const uint64_t*>(dp.mat)[dp_cell(R, (uint32_t)r, (uint32_t)c)]; v = (uint32_t)x; vh = (uint32_t)(x >> 32); }
					else v = dp.mat[dp_cell(R, (uint32_t)r, (uint32_t)c)];
				} else v = dp.masks[(uint64_t)r * cols + (uint32_t)c];
			}
			lo.v[ln] = v; hi.v[ln] = vh;
		}
	}
	static void load_seed_hits(const bt2g_seed_hit* src_fw, const bt2g_seed_hit* src_rc, uint32_t nseeds, bool skip_fw, bool skip_rc) {
		for (int fwi = 0; fwi < 2; fwi++) {
			const bt2g_seed_hit* src = fwi == 0 ? src_fw : src_rc;
			const bool skip = fwi == 0 ? skip_fw : skip_rc;
			for (uint32_t i = 0; i < nseeds; i++) {
				HotHit& h = g_hot.hits[fwi][i];
				g_hot.sorted[fwi][i] = 0;
				h.topf = h.topb = 0; h.size = h.esize = 0;
				if (skip) continue;
				const bt2g_seed_hit sh = src[i];
				if (sh.botf > sh.topf) { h.topf = sh.topf; h.topb = sh.topb; h.size = h.esize = (uint32_t)(sh.botf - sh.topf); }
			}
		}
	}
	static uint64_t seed_key(bool fw, uint32_t depth, uint32_t L, bool& ok) {
		uint64_t key = 0;
		ok = true;
		for (uint32_t k = 0; k < L; k++) {
			const int ch = fw ? (int)g_hot.seq[depth + k] : comp4(g_hot.seq[depth + L - 1 - k]);
			if (ch > 3) ok = false;      // (the key is still formed, with the N as 0: seed_round_mm1 substitutes that position)
			key = (key << 2) | (uint64_t)(ch & 3);
		}
		return key;
	}
	static uint32_t find_key(const uint64_t* keys, const uint8_t* lens, uint32_t n, uint64_t key, uint8_t len) {
		uint32_t e = 0;
		while (e < n && !(keys[e] == key && lens[e] == len)) e++;
		return e;
	}
	template <typename TOff>
	static void joined_to_text(const DevIndex<TOff>& ix, TOff qlen, TOff off, TOff& tidx, TOff& textoff, TOff& tlen, bool reject_straddle, bool& straddled) {
		uint64_t frag[4] = {0, 0, 0, ~0ull};
		joined_to_text_off(ix, qlen, off, tidx, textoff, tlen, reject_straddle, straddled, frag);
		g_hot.frag_jlo = frag[0]; g_hot.frag_len = frag[1]; g_hot.frag_toff = frag[2]; g_hot.frag_tidx = frag[3];
	}
	static void fetch_ref_joined(const DevRef& ref, uint64_t jpos, uint32_t count) {
		for (uint32_t i = 0; i < count; i++) { const uint64_t p = jpos + i; host_rf[i] = (uint8_t)(1u << ((ref.buf[p >> 2] >> ((p & 3) << 1)) & 3)); }
	}
	static bool diag_find(const DiagIval* d, uint32_t n, int32_t ref, int64_t off, int32_t orient) {
		for (uint32_t i = 0; i < n; i++) if (d[i].ref == ref && d[i].orient == orient && off >= d[i].off && off < d[i].off + d[i].len) return true;
		return false;
	}
	static uint32_t bt_diag_run(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, const LaneReg& tile, const LaneReg& tile_hi,
	                            uint32_t td, uint32_t row, uint32_t col, bool fw, uint32_t rdlen, uint32_t maxl, LaneReg& info, uint64_t& mm) {
		mm = 0;
		uint32_t L = 0;
		for (uint32_t d = td; d < 64; d++, L++) {
			const uint32_t k = d - td;
			if (!(k < row && k <= col && k < maxl)) break;
			const uint32_t pb = tile.v[d];
			const bool he = (pb & PB_HE) != 0, hf = (pb & PB_HF) != 0;
			if (!(tile_hi.v[d] == 0 && (pb & PB_HD) && !(hf && (pb & (PB_FO | PB_FE))) && !(he && (pb & (PB_EO | PB_EE))))) break;
		}
		for (uint32_t k = 0; k < L; k++) {
			const uint32_t r = row - k, c = col - k;
			const int readc = rd_char(g_hot, rdlen, fw, r);
			const int refm = host_rf[c];
			const int readq = rd_qual(g_hot, rdlen, fw, r);
			const int m = (refm >= 16 || readc > 3) ? -1 : (((1 << readc) & refm) ? 1 : 0);
			if (m != 1) mm |= 1ull << (td + k);
			info.v[td + k] = ((uint32_t)readc << 4) | ((uint32_t)refm << 8) | ((uint32_t)readq << 16) | (m == -1 ? 2u : 0u);
			dp.pmask[pred_at(band_lo, band_w, r, c)] = 1u | (epoch << kEpochShift);
		}
		return L;
	}
	static uint32_t bt_gap_run(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, const LaneReg& tile, const LaneReg& tile_hi, uint32_t td, uint32_t row,
	                           uint32_t col, bool read_gap, bool fw, uint32_t rdlen, uint32_t maxl, uint32_t nned, int r_triml, int r_corel, int r_corer, uint32_t& core) {
		core = 0;
		uint32_t L = 0;
		for (uint32_t d = td; d < 64; d++, L++) {
			const uint32_t k = d - td;
			if (!(k < maxl && (read_gap ? k <= col : k < row))) break;
			const uint32_t pb = tile.v[d];
			const uint32_t m = read_gap ? (pb >> 3) & 3u : (pb >> 5) & 3u;
			if (!(tile_hi.v[d] == 0 && m == 2u)) break;
		}
		for (uint32_t k = 0; k < L; k++) {
			const uint32_t r = read_gap ? row : row - k, c = read_gap ? col - k : col;
			Edit e;
			if (read_gap) { const int refm = host_rf[c]; e.pos = (uint16_t)(r + 1); e.chr = (uint8_t)((refm == 1 || refm == 2 || refm == 4 || refm == 8) ? code2chr(__builtin_ctz((unsigned)refm)) : 'N'); e.qchr = '-'; e.type = EDIT_READ_GAP; }
			else { e.pos = (uint16_t)r; e.chr = '-'; e.qchr = code2chr(rd_char(g_hot, rdlen, fw, r)); e.type = EDIT_REF_GAP; }
			e.pad = 0;
			host_ned[nned + k] = e;
			dp.pmask[pred_at(band_lo, band_w, r, c)] = 1u | (epoch << kEpochShift);
			const int diagi = (int)c - (int)r + r_triml;
			if (diagi >= r_corel && diagi <= r_corer) core = 1;
		}
		return L;
	}
	static void rt_begin(const DpScratch&, uint32_t, bool) {}
	static void rt_mark(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t row, uint32_t col) { dp.pmask[pred_at(band_lo, band_w, row, col)] = 1u | (epoch << kEpochShift); }
	// ---- the candidates that die within a few cells, side by side (Aligner::next_alignment_m): per-lane predecessor bytes, one-cell mark operations ----
	static bool marks_batchable() { static const bool off = getenv("BT2G_HOST_NO_WALK_BATCH") != nullptr; return !off; }
	static bool cell_ok(int32_t band_lo, uint32_t band_w, uint32_t r, uint32_t c) {
		return (int32_t)r >= 0 && (int32_t)c >= 0 && (band_w == 0u || (uint32_t)((int32_t)c - (int32_t)r + band_lo) < band_w);
	}
	static bool mark_of(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, uint32_t r, uint32_t c) {
		if (!cell_ok(band_lo, band_w, r, c)) return false;
		const uint32_t w = dp.pmask[pred_at(band_lo, band_w, r, c)];
		return (w >> kEpochShift) == epoch && (w & 1u);
	}
	// per lane: the predecessor bytes (plo: cells 0-3, phi: cells 4-7) and reportedThrough bits (mk, bit i) of the eight cells from the lane's own cell on in
	// its own direction (0 up the diagonal, 1 left along the row, 2 up the column); a cell outside the band reads as 0 / unmarked
	static void pred_tile8(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, const LaneReg& row, const LaneReg& col, const LaneReg& dir, const LaneReg& flag,
	                       LaneReg& plo, LaneReg& phi, LaneReg& mk) {
		for (uint32_t l = 0; l < 64; l++) {
			plo.v[l] = phi.v[l] = mk.v[l] = 0;
			if (!flag.v[l]) continue;
			for (uint32_t i = 0; i < 8; i++) {
				const uint32_t r = row.v[l] - (dir.v[l] == 1u ? 0u : i), c = col.v[l] - (dir.v[l] == 2u ? 0u : i);
				if (!cell_ok(band_lo, band_w, r, c)) continue;
				const uint32_t p = reinterpret_cast<const uint8_t*>(dp.mat)[pred_at(band_lo, band_w, r, c)];
				if (i < 4) plo.v[l] |= p << (8 * i); else phi.v[l] |= p << (8 * (i - 4));
				if (mark_of(dp, band_lo, band_w, epoch, r, c)) mk.v[l] |= 1u << i;
			}
		}
	}
	// per lane: reportedThrough of the lane's own cell / set it
	static LaneReg marks_of_cells(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, const LaneReg& row, const LaneReg& col, const LaneReg& flag) {
		LaneReg r;
		for (uint32_t l = 0; l < 64; l++) r.v[l] = (flag.v[l] && mark_of(dp, band_lo, band_w, epoch, row.v[l], col.v[l])) ? 1u : 0u;
		return r;
	}
	static void mark_cells(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t epoch, const LaneReg& row, const LaneReg& col, const LaneReg& flag) {
		for (uint32_t l = 0; l < 64; l++) if (flag.v[l] && cell_ok(band_lo, band_w, row.v[l], col.v[l])) dp.pmask[pred_at(band_lo, band_w, row.v[l], col.v[l])] = 1u | (epoch << kEpochShift);
	}
	static void bt_tile_pred(const DpScratch& dp, int32_t band_lo, uint32_t band_w, uint32_t row, uint32_t col, uint32_t epoch, uint32_t dir, LaneReg& pr, LaneReg& mk) {
		for (uint32_t d = 0; d < 64; d++) {
			uint32_t p = 0, m = 0;
			const uint32_t dr = dir == 1 ? 0u : d, dc = dir == 2 ? 0u : d;
			const uint32_t dd = (uint32_t)((int32_t)col - (int32_t)row + band_lo) + dr - dc;
			if (dr <= row && dc <= col && (band_w == 0u || dd < band_w)) {
				const uint64_t idx = pred_at(band_lo, band_w, row - dr, col - dc);
				p = reinterpret_cast<const uint8_t*>(dp.mat)[idx];
				const uint32_t w = dp.pmask[idx];
				m = (w >> kEpochShift) == epoch ? (w & 1u) : 0u;
			}
			pr.v[d] = p; mk.v[d] = m;
		}
	}
	// scalar fill of the reference recurrence into the wavefront-major layout; returns best last-row H
	// Local-mode fill (alignNucleotidesLocalSseU8 / ...I16 agree wherever the 8-bit kernel does not saturate): plain
	// scores in 16-bit fields, floor 0.  Returns the best score; lastsolcol = last column whose maximum reaches minsc;
	// sat8 = the 8-bit kernel would have saturated (some column maximum + bias >= 255 before its bail-out point).
	static int64_t dp_fill_local(const AlignParams& P, Work& w, bool fw, uint32_t rows, uint32_t cols, uint32_t* mat, int64_t minsc,
	                             uint32_t& lastsolcol, uint32_t& sat8) {
		const uint32_t R = dp_RB(rows);
		uint8_t* pm = reinterpret_cast<uint8_t*>(mat);      // one predecessor byte per cell, anti-diagonal order (pred_at with w = 0, lo = rows per block)
		st().dp.epoch[1] = R; st().dp.epoch[2] = 0u;
		static const bool check_pk = getenv("BT2G_CHECK_LOCAL_PK") != nullptr;
		if (check_pk) local_ef().assign((size_t)rows * cols, 0);
		local_h().assign((size_t)rows * cols, 0);          // the scores themselves, for gather_local (the device's fill emits the candidates instead)
		auto subs = [](int a, int b) { const int v = a - b; return v < 0 ? 0 : v; };
		// bias of the 8-bit query profile: the largest penalty any (read position, reference character) pair can incur
		int bias = P.n_pen;
		for (uint32_t i = 0; i < rows; i++) {
			const int rdc = rd_char(g_hot, g_hot.len, fw, i);
			const int q = rd_qual(g_hot, g_hot.len, fw, i) - 33;
			if (rdc <= 3) { const int mp = mm_penalty(P, q < 0 ? 0 : q); if (mp > bias) bias = mp; }
		}
		std::vector<int> Hp(rows, 0), Ep(rows, 0), Hc(rows), Ec(rows), Fc(rows);
		int vmax = 0;
		bool bailed = false;
		lastsolcol = 0; sat8 = 0;
		for (uint32_t j = 0; j < cols; j++) {
			const int m = host_rf[j];
			int refc = 4;
			for (int b = 0; b < 4; b++) if (m & (1 << b)) { refc = b; break; }
			int f = 0, colmax = 0;
			for (uint32_t i = 0; i < rows; i++) {
				const bool veto = ((int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar);
				const int rdc = rd_char(g_hot, g_hot.len, fw, i);
				const int q = rd_qual(g_hot, g_hot.len, fw, i) - 33;
				int sc;
				if (rdc > 3 || refc > 3) sc = -P.n_pen; else sc = (rdc == refc) ? P.match_bonus : -mm_penalty(P, q < 0 ? 0 : q);
				const int hdiag = (i == 0 || j == 0) ? 0 : Hp[i - 1];
				const int hl = (j == 0) ? 0 : Hp[i], el = (j == 0) ? 0 : Ep[i];
				const int hu = (i == 0) ? 0 : Hc[i - 1], fu = (i == 0) ? 0 : f;
				const int e = (j == 0) ? 0 : imax(subs(el, P.rdgape), veto ? 0 : subs(hl, P.rdgapo));
				f = (i == 0) ? 0 : (veto ? 0 : imax(subs(fu, P.rfgape), subs(hu, P.rfgapo)));
				int h = hdiag + sc; if (h < 0) h = 0;
				h = imax(imax(h, e), f);
				Hc[i] = h; Ec[i] = e; Fc[i] = f;
				if (h > colmax) colmax = h;
				// predecessor bits with the local kernels' `> floor` rule (aligner_swsse_loc_u8.cpp:1530-1660): a neighbour whose score is 0 is none
				int c = (hdiag > 0 && hdiag + sc == h) ? PB_HD : 0;
				c |= (!veto && h == e) ? PB_HE : 0;
				c |= (!veto && h == f) ? PB_HF : 0;
				c |= (hl > 0 && hl - P.rdgapo == e) ? PB_EO : 0;
				c |= (el > 0 && el - P.rdgape == e) ? PB_EE : 0;
				c |= (hu > 0 && hu - P.rfgapo == f) ? PB_FO : 0;
				c |= (fu > 0 && fu - P.rfgape == f) ? PB_FE : 0;
				pm[dp_cell_pk(R, i, j)] = (uint8_t)c;
				if (check_pk) local_ef()[(size_t)i * cols + j] = (uint8_t)((e > 0 ? 1 : 0) | (f > 0 ? 2 : 0));
				local_h()[(size_t)i * cols + j] = (uint16_t)h;
			}
			if (!bailed) {
				if (colmax > vmax) vmax = colmax;
				if (colmax + bias >= 255) sat8 = 1;
				if (colmax < minsc) {
					if ((int64_t)colmax + (int64_t)(cols - j - 1) * P.match_bonus < minsc) bailed = true;   // the kernels stop here
				} else lastsolcol = j;
			}
			Hp.swap(Hc); Ep.swap(Ec);
		}
		if (check_pk) check_local_pk(P, fw, rows, cols, pm, minsc, vmax, lastsolcol, sat8, bias);
		return (int64_t)vmax;
	}
	// BT2G_CHECK_LOCAL_PK=1: the device's local fill computes two cells per register (bt2g_local_pk.hpp) and moves data between lanes; this
	// replays it -- 64 lanes side by side, the same per-lane step function, the same lane-to-lane hand-over and per-column bookkeeping as
	// fill_local_pk (bt2g_align_kernel.hip) -- on the window just filled and compares every predecessor byte (the bits somebody can look at:
	// H bits where H > 0, E bits where E > 0, F bits where F > 0) and the fill's results with the scalar fill's.  Aborts on a difference.
	template <int RB>
	static void check_local_pk_t(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols, const uint8_t* pm, int64_t minsc, int want_best, uint32_t want_lastsol, uint32_t want_sat, int bias) {
		const uint32_t nblocks = (rows + RB - 1) / RB;
		LocalPkRows<RB> K[64];
		for (uint32_t lane = 0; lane < 64; lane++)
			for (int r = 0; r < RB; r++) {
				uint32_t km = 0, kb = 0, kn = 0, kv = 0, ko = 0;
				for (int half = 0; half < 2; half++) {
					const uint32_t i = (lane + 64u * half) * RB + r;
					const bool valid = i < rows;
					const int rdc = valid ? rd_char(g_hot, g_hot.len, fw, i) : 4;
					const int q = valid ? rd_qual(g_hot, g_hot.len, fw, i) - 33 : 0;
					const int mmp = mm_penalty(P, q < 0 ? 0 : q);
					const bool veto = (int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar;
					const LocalPkRow1 c = local_pk_row(valid, rdc, mmp, veto, P.match_bonus, P.n_pen);
					km |= c.rowmask << (16 * half); kb |= c.bpm << (16 * half); kn |= c.nmmp << (16 * half); kv |= c.vm << (16 * half); ko |= c.okm << (16 * half);
				}
				K[lane].rowmask[r] = km; K[lane].bpm[r] = kb; K[lane].nmmp[r] = kn; K[lane].vm[r] = kv; K[lane].okm[r] = ko;
			}
		const LocalPkPen G = local_pk_pen(P.rdgapo, P.rdgape, P.rfgapo, P.rfgape, P.n_pen);
		uint32_t Hp[64][RB] = {}, Ep[64][RB] = {};
		uint32_t myH[64] = {}, myF[64] = {}, mycm[64] = {}, upHdiag[64] = {}, refm[64];
		for (auto& x : refm) x = 0x00100010u;
		int vmax = 0, lastsol = 0, sat = 0, bailed = 0;
		const uint32_t last_blk = nblocks - 1, last_lane = last_blk & 63u, last_half = last_blk >> 6;
		const uint32_t steps = cols + nblocks - 1;
		const int ms = minsc > 0x7fff ? 0x7fff : (int)minsc;
		uint64_t nbad = 0;
		for (uint32_t t = 0; t < steps; t++) {
			uint32_t upH[64], upF[64], upM[64], upR[64];
			for (uint32_t l = 1; l < 64; l++) { upH[l] = myH[l - 1]; upF[l] = myF[l - 1]; upM[l] = mycm[l - 1]; upR[l] = refm[l - 1]; }
			upH[0] = myH[63] << 16; upF[0] = myF[63] << 16; upM[0] = mycm[63] << 16; upR[0] = (refm[63] << 16) | (t < cols ? (uint32_t)host_rf[t] : 16u);
			uint32_t cmv[64];
			for (uint32_t l = 0; l < 64; l++) {
				refm[l] = upR[l];
				uint32_t pb[RB];
				uint32_t cm = upM[l];
				local_pk_step<RB>(K[l], G, refm[l], upHdiag[l], upH[l], upF[l], Hp[l], Ep[l], pb, myH[l], myF[l], cm);
				mycm[l] = cm; cmv[l] = cm;
				upHdiag[l] = upH[l];
				for (int half = 0; half < 2; half++) {
					const uint32_t k = l + 64u * half, j = t - k;
					if (k >= nblocks || j >= cols) continue;
					for (int r = 0; r < RB; r++) {
						const uint32_t i = k * RB + r;
						if (i >= rows) continue;
						const uint32_t got = (pb[r] >> (16 * half)) & 0xffu, want = pm[dp_cell_pk(RB, i, j)];
						const uint32_t h = local_h()[(size_t)i * cols + j];
						const uint32_t hgot = (Hp[l][r] >> (16 * half)) & 0xffffu, egot = (Ep[l][r] >> (16 * half)) & 0xffffu;
												const uint32_t ef = local_ef()[(size_t)i * cols + j];
						const uint32_t care = (h > 0 ? 7u : 0u) | ((ef & 1u) ? 24u : 0u) | ((ef & 2u) ? 96u : 0u);
						if ((egot > 0) != ((ef & 1u) != 0)) nbad++;
						if (hgot != h || ((got ^ want) & care)) {
							if (nbad++ < 5) fprintf(stderr, "local pk check: cell (%u, %u) of %u x %u: H %u vs %u, pred %02x vs %02x (care %02x)\n", i, j, rows, cols, hgot, h, got, want, care);
						}
					}
				}
			}
			if (t >= last_blk && !bailed) {
				const int j = (int)(t - last_blk);
				const int c = (int)(last_half ? cmv[last_lane] >> 16 : cmv[last_lane] & 0xffffu);
				if (c > vmax) vmax = c;
				if (c + bias >= 255) sat = 1;
				if (c < ms) { if (c + (int)(cols - (uint32_t)j - 1) * P.match_bonus < ms) bailed = 1; }
				else lastsol = j;
			}
		}
		if (vmax != want_best || (uint32_t)lastsol != want_lastsol || (uint32_t)sat != want_sat) { fprintf(stderr, "local pk check: best %d vs %d, lastsol %d vs %u, sat %d vs %u\n", vmax, want_best, lastsol, want_lastsol, sat, want_sat); nbad++; }
		if (nbad) { fprintf(stderr, "local pk check: %llu differences (%u x %u, RB %d)\n", (unsigned long long)nbad, rows, cols, RB); abort(); }
		static unsigned long n_checked = 0;
		if (++n_checked == 1) fprintf(stderr, "[hostsim] local pk check active\n");
	}
	static void check_local_pk(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols, const uint8_t* pm, int64_t minsc, int want_best, uint32_t want_lastsol, uint32_t want_sat, int bias) {
		switch (dp_RB(rows)) {
			case 1: check_local_pk_t<1>(P, fw, rows, cols, pm, minsc, want_best, want_lastsol, want_sat, bias); break;
			case 2: check_local_pk_t<2>(P, fw, rows, cols, pm, minsc, want_best, want_lastsol, want_sat, bias); break;
			case 3: check_local_pk_t<3>(P, fw, rows, cols, pm, minsc, want_best, want_lastsol, want_sat, bias); break;
			default: check_local_pk_t<4>(P, fw, rows, cols, pm, minsc, want_best, want_lastsol, want_sat, bias); break;
		}
	}
	// Candidate cells of a local fill, sorted score desc, row desc, col desc
	static uint32_t gather_local(const uint32_t* mat, BtCand* cands, uint32_t cap, bool fw, uint32_t R, uint32_t rows, uint32_t ncol,
	                             int64_t minsc, uint32_t minrow, uint32_t* hist) {
		memset(hist, 0xa5, sizeof(uint32_t) * 2 * (kMaxLocalScore + 1));      // the device's counting sort scribbles over its scratch: nothing may live there across a gather
		const size_t hcols = local_h().size() / (rows ? rows : 1);
		uint32_t n = 0, total = 0;
		for (uint32_t j = 0; j < ncol; j++) {
			for (uint32_t i = minrow; i < rows; i++) {
				const int sc = (int)local_h()[(size_t)i * hcols + j];
				if (sc < minsc) continue;
				const int rdc = rd_char(g_hot, g_hot.len, fw, i);
				const bool match = (host_rf[j] & (1 << rdc)) != 0;        // as the reference: a read N "matches" a reference N mask (16)
				bool match_succ = false;
				if (i < rows - 1) { const int rs = rd_char(g_hot, g_hot.len, fw, i + 1); match_succ = (host_rf[j + 1] & (1 << rs)) != 0; }
				if (!match || match_succ) continue;
				total++;
				if (n >= cap) continue;
				BtCand c; c.score = sc; c.row = (uint16_t)i; c.col = (uint16_t)j;
				cands[n++] = c;
			}
		}
		std::sort(cands, cands + n, [](const BtCand& a, const BtCand& b) {      // score desc, row desc, col desc
			if (a.score != b.score) return a.score > b.score;
			if (a.row != b.row) return a.row > b.row;
			return a.col > b.col;
		});
		return total;
	}
	// 8-bit end-to-end, pred format: the band of diagonals a valid alignment can touch (EeBand), everything outside it "minus
	// infinity" (0) -- the exactness argument is in bt2g_align.hpp; this scalar form is what the differential tests pin it with
	static int64_t fill_ee_u8_band(const AlignParams& P, bool fw, uint32_t rows, uint32_t cols, const DpScratch& dp, int64_t minsc) {
		EeBand band;
		if (!ee_band(P.rfgapo, P.rfgape, rows, cols, minsc, band)) return -0xff;
		// test hook (tests/test_band_fill.py): fill every diagonal of the rectangle instead of the band -- the two must agree on
		// everything the worker can observe
		static const bool full_rect = getenv("BT2G_HOST_FULL_RECT") != nullptr;
		if (full_rect) { band.lo = (int32_t)rows - 1; band.nd = rows + cols - 1; }
		const uint32_t rp = ee_band_rp(band.nd);
		if (rp == 0) return INT64_MIN;
		const int32_t lo = band.lo;
		const uint32_t W = 128u * rp;
		dp.epoch[1] = (uint32_t)lo; dp.epoch[2] = W;
		uint8_t* pm = reinterpret_cast<uint8_t*>(dp.mat);
		auto subs = [](int a, int b) { const int v = a - b; return v < 0 ? 0 : v; };
		// previous / current row, indexed by column + 1 (index 0 = the column left of the window)
		std::vector<int> Hp(cols + 2, 0), Fp(cols + 2, 0), Hc(cols + 2, 0), Ec(cols + 2, 0), Fc(cols + 2, 0);
		for (uint32_t j = 0; j < cols; j++) host_lastrow[j] = (int16_t)-0xff;
		int lrmax = 0;
		for (uint32_t i = 0; i < rows; i++) {
			const bool veto = ((int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar);
			const int rdc = rd_char(g_hot, g_hot.len, fw, i);
			const int q = rd_qual(g_hot, g_hot.len, fw, i) - 33;
			std::fill(Hc.begin(), Hc.end(), 0); std::fill(Ec.begin(), Ec.end(), 0); std::fill(Fc.begin(), Fc.end(), 0);
			// columns of this row inside the band
			const int64_t jlo = (int64_t)i - lo, jhi = (int64_t)i - lo + (int64_t)band.nd - 1;
			for (int64_t jj = jlo < 0 ? 0 : jlo; jj <= jhi && jj < (int64_t)cols; jj++) {
				const uint32_t j = (uint32_t)jj;
				const int m = host_rf[j];
				int refc = 4;
				for (int b = 0; b < 4; b++) if (m & (1 << b)) { refc = b; break; }
				int pen;
				if (rdc > 3 || refc > 3) pen = P.n_pen; else pen = (rdc == refc) ? 0 : mm_penalty(P, q < 0 ? 0 : q);
				const int hdiag = (i == 0) ? 0xff : Hp[j];          // Hp[j] = H(i-1, j-1); column -1 and cells outside the band hold 0
				const int hl = Hc[j], el = Ec[j];                   // (i, j-1)
				const int hu = (i == 0) ? 0 : Hp[j + 1], fu = (i == 0) ? 0 : Fp[j + 1];
				const int e = veto ? 0 : imax(subs(el, P.rdgape), subs(hl, P.rdgapo));
				const int f = (i == 0 || veto) ? 0 : imax(subs(fu, P.rfgape), subs(hu, P.rfgapo));
				const int h = imax(imax(subs(hdiag, pen), e), f);
				const bool ga = !veto;
				int c = (hdiag - pen == h) ? PB_HD : 0;
				c |= (ga && h == e) ? PB_HE : 0;
				c |= (ga && h == f) ? PB_HF : 0;
				c |= (hl - P.rdgapo == e) ? PB_EO : 0;
				c |= (el - P.rdgape == e) ? PB_EE : 0;
				c |= (hu - P.rfgapo == f) ? PB_FO : 0;
				c |= (fu - P.rfgape == f) ? PB_FE : 0;
				pm[pred_idx(lo, W, i, j)] = (uint8_t)c;
				Hc[j + 1] = h; Ec[j + 1] = e; Fc[j + 1] = f;
				if (i == rows - 1) { host_lastrow[j] = (int16_t)(h - 0xff); if (h > lrmax) lrmax = h; }
			}
			Hp.swap(Hc); Fp.swap(Fc);
		}
		return (int64_t)lrmax - 0xff;
	}
	// scalar fill of either representation; returns the best last-row score (de-biased)
	static int64_t dp_fill_ee(const AlignParams& P, Work& w, bool fw, uint32_t rows, uint32_t cols, const DpScratch& dp, bool wide, int64_t minsc) {
		if (!wide) return fill_ee_u8_band(P, fw, rows, cols, dp, minsc);
		uint32_t* mat = dp.mat;
		const uint32_t R = dp_R(rows);
		const int lo = -32768, hi = 0x7fff;
		auto subs = [&](int a, int b) { const int v = a - b; return v < lo ? lo : v; };
		int lrmax = lo;
		std::vector<int> Hp(rows, lo), Ep(rows, lo), Hc(rows), Ec(rows), Fc(rows);
		uint64_t* m64 = reinterpret_cast<uint64_t*>(mat);
		for (uint32_t j = 0; j < cols; j++) {
			const int m = host_rf[j];
			int refc = 4;
			for (int b = 0; b < 4; b++) if (m & (1 << b)) { refc = b; break; }
			int f = lo;
			for (uint32_t i = 0; i < rows; i++) {
				const bool veto = ((int)i < P.gapbar || (int)(rows - i - 1) < P.gapbar);
				const int rdc = rd_char(g_hot, g_hot.len, fw, i);
				const int q = rd_qual(g_hot, g_hot.len, fw, i) - 33;
				int pen;
				if (rdc > 3 || refc > 3) pen = P.n_pen; else pen = (rdc == refc) ? -P.match_bonus : mm_penalty(P, q < 0 ? 0 : q);
				const int hdiag = (i == 0) ? hi : (j == 0 ? lo : Hp[i - 1]);
				const int e = (j == 0) ? lo : imax(subs(Ep[i], P.rdgape), veto ? lo : subs(Hp[i], P.rdgapo));
				f = (i == 0) ? lo : (veto ? lo : imax(subs(f, P.rfgape), subs(Hc[i - 1], P.rfgapo)));
				const int h = imax(imax(subs(hdiag, pen), e), f);
				m64[dp_cell(R, i, j)] = (uint64_t)(uint16_t)h | ((uint64_t)(uint16_t)e << 16) | ((uint64_t)(uint16_t)f << 32);
				Hc[i] = h; Ec[i] = e; Fc[i] = f;
			}
			if (Hc[rows - 1] > lrmax) lrmax = Hc[rows - 1];
			Hp.swap(Hc); Ep.swap(Ec);
		}
		return (int64_t)lrmax - hi;
	}
};

template <typename TOff>
static void make_dev_index(const HostIndex& h, DevIndex<TOff>& d) {
	// the layout bt2g_index_load builds on the device (rank blocks, full suffix array), built here by the same per-block /
	// per-segment functions in plain loops (bt2g_rankidx.hpp); the buffers live as long as the process
	auto fill = [&](const HostEbwt& e, DevEbwt<TOff>& o, bool fw) {
		o.ftab = (const TOff*)e.ftab.data();
		o.eftab = (const TOff*)e.eftab.data();
		o.sa = nullptr;
		o.len = (TOff)e.len; o.zoff = (TOff)e.zoff;
		o.zblk = (uint64_t)e.zoff >> kBlkShift; o.zchar = (uint32_t)((uint64_t)e.zoff & (kBlkLen - 1));
		for (int i = 0; i < 5; i++) o.fchr[i] = (TOff)e.fchr[i];
		o.ftab_chars = e.ftab_chars; o.off_rate = e.off_rate; o.is_fw = fw;
		const uint64_t n_sides = e.ebwt.size() / OffTraits<TOff>::kSideSz;
		const uint64_t n_blocks = rank_block_count(n_sides, OffTraits<TOff>::kSideBwtLen);
		RankBlock* blk = (RankBlock*)aligned_alloc(64, n_blocks * sizeof(RankBlock));
		host_make_rank_blocks<TOff>(e.ebwt.data(), n_sides, o.fchr, o.zoff, blk, n_blocks);
		o.blk = blk;
		if (fw) {
			uint64_t* sa = (uint64_t*)malloc(((uint64_t)e.len + 1) * sizeof(uint64_t));
			for (uint64_t r = 0; r <= (uint64_t)e.len; r++) sa[r] = kJoffNone;
			const uint64_t lost = host_make_full_sa_checked<TOff>(o, (const TOff*)e.offs.data(), sa);
			if (lost) { fprintf(stderr, "Error: the suffix-array sample of this index (--offrate %d) leaves %llu rows more than 65534 LF steps from a sampled row (not supported)\n", (int)e.off_rate, (unsigned long long)lost); exit(1); }      // as bt2g_index_load
			o.sa = sa;
		}
	};
	fill(h.fw, d.fw, true);
	fill(h.bw, d.bw, false);
	d.rstarts = (const TOff*)h.fw.rstarts.data();
	d.plen = (const TOff*)h.fw.plen.data();
	d.n_frag = (TOff)h.fw.n_frag; d.n_pat = (TOff)h.fw.n_pat;
	d.ref.rec_refpos = h.ref.rec_refpos.data(); d.ref.rec_bufpos = h.ref.rec_bufpos.data(); d.ref.rec_len = h.ref.rec_len.data();
	d.ref.ref_rec_offs = h.ref.ref_rec_offs.data(); d.ref.ref_lens = h.ref.ref_lens.data(); d.ref.buf = h.ref.buf.data();
	d.ref.nrefs = h.ref.nrefs;
}

template <typename TOff>
static int run(const HostIndex& hidx, const Options& opt, FILE* out, bool metrics, bool header = true, AlnSummary* keep = nullptr) {
	DevIndex<TOff> ix;
	make_dev_index(hidx, ix);
	AlignParams P;
	opt.to_params(P, sizeof(TOff) == 8);
	RefInfo ref;
	ref.names = hidx.fw.refnames;
	for (uint64_t i = 0; i < hidx.fw.n_pat; i++) ref.lens.push_back(hidx.plen_at(i));
	std::string o;
	if (!opt.sam_no_hd && header) sam_header(o, ref, opt.cmdline, true, !opt.sam_no_sq, opt.rg_id, opt.rgs);   // --no-hd drops every header line (bt2_search.cpp:5126-5130)
	fwrite(o.data(), 1, o.size(), out);
	FastqBatcher fq(opt.reads_file, opt, 1);       // the product's reader, single-threaded
	if (!fq.ok()) { fprintf(stderr, "cannot open %s\n", opt.reads_file.c_str()); return 1; }
	Work* w = (Work*)calloc(1, sizeof(Work));      // (zero pages on first touch: value-initialising the class-sized Work cost every start of the twin ~50 ms)
	DpScratch dp;
	const uint64_t mat_bytes = ((uint64_t)kMaxColsWide + 64) * dp_R(kMaxLen) * 64 * 8;
	dp.mat = (uint32_t*)malloc(mat_bytes);
	dp.masks = (uint16_t*)malloc((size_t)kMaxLen * (kMaxColsWide + 8) * 2);
	dp.pmask_words = (uint32_t)pred_cells(kMaxLen, kMaxColsWide + 8);
	dp.pmask = (uint32_t*)calloc(dp.pmask_words, 4); dp.epoch = (uint32_t*)calloc(64, 4);
	std::vector<uint8_t> resbuf(sizeof(ReadResult) + sizeof(AlnRes) * (size_t)(P.khits + 1));
	AlnSummary summ;
	HostBatch hb;
	// --shard r/N as in the product binary: batch k = block k of the input, this rank keeps blocks r, r+N, ...
	FILE* shard_idx = g_ex.shard_index.empty() ? nullptr : fopen(g_ex.shard_index.c_str(), "w");
	uint64_t blk = 0;
	for (bool last = false; !last; ) {
	hb = HostBatch();
	fq.next(hb, std::min<size_t>(g_ex.batch_reads, 4096), (size_t)1 << 30);
	last = hb.last;
	if (!hb.bad_input.empty()) { fprintf(stderr, "Error: %s\n", hb.bad_input.c_str()); return 1; }
	const uint64_t block_id = blk++;
	if (g_ex.shard_world > 1 && (int)(block_id % (uint64_t)g_ex.shard_world) != g_ex.shard_rank) continue;
	uint64_t nbytes = 0;
	for (size_t ri = 0; ri < hb.reads.size(); ri++) {
		const ReadRec& rd = hb.reads[ri];
		ReadResult& rr = *(ReadResult*)resbuf.data();
		if (rd.seq.size() > (size_t)BT2G_MAX_READ_LEN || rd.seq.size() > (size_t)kMaxLen) {
			fprintf(stderr, "Error: read %s is longer than %d bp (unsupported)\n", rd.name.str().c_str(), kMaxLen < BT2G_MAX_READ_LEN ? kMaxLen : BT2G_MAX_READ_LEN);
			return 1;
		}
		g_rp = hb.rp[ri]; g_Pp = &P; g_ixp = &ix;
		g_hot.len = (uint32_t)rd.seq.size();
		memcpy(g_hot.seq, rd.seq.data(), rd.seq.size());
		memcpy(g_hot.qual, rd.qual.data(), rd.qual.size());
		g_st.max_cols = (uint32_t)kMaxColsWide;      // (the product's driver asks bt2g_align_batch for what its pairs need, up to this: bt2g_align_params::max_dp_cols)
		Aligner<TOff, HostPlat> al(*w, dp);
		al.run(rr);
		if (rr.status) fprintf(stderr, "Warning: read %s overflowed a fixed-capacity buffer (status %d, site %u)\n", rd.name.str().c_str(), rr.status, rr.pad2 & 0xffffu);
		summ.add(rr);
		o.clear();
		if (rr.aligned) {
			for (uint32_t i = 0; i < rr.nreport; i++) sam_record(o, opt, ref, rd, rr, &rr.alns[i], i == 0);
		} else if (!opt.no_unal) {
			sam_record(o, opt, ref, rd, rr, nullptr, true);
		}
		fwrite(o.data(), 1, o.size(), out);
		nbytes += o.size();
		if (metrics) fprintf(stderr, "MET\t%s\titers=%u dps=%u ugs=%u bwseed=%u bwext=%u red=%u bt=%u nalns=%u extl=%u extr=%u res=%u btsteps=%llu tiles=%llu cands=%llu draws=%llu batched=%llu batches=%u\n", rd.name.str().c_str(),
		                     rr.n_ex_iters, rr.n_ex_dps, rr.n_ex_ugs, rr.n_bwops_seed, rr.n_bwops_ext, rr.n_redundants, rr.n_bt_attempts, rr.nalns, rr.n_ext_left, rr.n_ext_right, rr.n_resolve_steps,
		                     (unsigned long long)g_hot.t_phase[11], (unsigned long long)g_hot.t_phase[12], (unsigned long long)g_hot.t_phase[13],
		                     (unsigned long long)(g_hot.t_phase[21] & 0xffffffffull), (unsigned long long)(g_hot.t_phase[21] >> 32), (unsigned)(g_hot.n_dp_pass >> 16));
	}
	if (shard_idx && !hb.reads.empty()) fprintf(shard_idx, "B %llu %llu %llu\n", (unsigned long long)block_id, (unsigned long long)nbytes, (unsigned long long)hb.reads.size());
	}
	if (shard_idx) {
		fprintf(shard_idx, "S %llu %llu %llu %llu\nP 0 0 0 0 0 0 0 0 0 0\nF 0\n", (unsigned long long)summ.nread, (unsigned long long)summ.n0, (unsigned long long)summ.nuni, (unsigned long long)summ.nrep);
		fclose(shard_idx);
	}
	if (keep) *keep = summ;
	else if (!opt.quiet && g_ex.shard_world == 1) summ.print(stderr);
	return 0;
}

// paired-end flavour of run(): two single-mate readers, one pair at a time through Aligner::run_pair
template <typename TOff>
static int run_pairs(const HostIndex& hidx, const Options& opt, FILE* out, bool metrics, PairSummary* keep = nullptr, bool* upto_hit = nullptr) {
	DevIndex<TOff> ix;
	make_dev_index(hidx, ix);
	AlignParams P;
	opt.to_params(P, sizeof(TOff) == 8);
	RefInfo ref;
	ref.names = hidx.fw.refnames;
	for (uint64_t i = 0; i < hidx.fw.n_pat; i++) ref.lens.push_back(hidx.plen_at(i));
	std::string o;
	if (!opt.sam_no_hd) sam_header(o, ref, opt.cmdline, true, !opt.sam_no_sq, opt.rg_id, opt.rgs);
	fwrite(o.data(), 1, o.size(), out);
	const bool inter = !opt.interleaved_file.empty();
	FastqBatcher fq1(inter ? opt.interleaved_file : opt.mate1_file, opt, 1), fq2(inter ? std::string("/dev/null") : opt.mate2_file, opt, 1);
	if (!inter) { fq1.set_bam_mate(1); fq2.set_bam_mate(2); }
	if (!fq1.ok() || !fq2.ok()) { fprintf(stderr, "cannot open the mate files\n"); return 1; }
	Work* w = (Work*)calloc(1, sizeof(Work));      // (zero pages on first touch: value-initialising the class-sized Work cost every start of the twin ~50 ms)
	DpScratch dp, dp2;
	const uint64_t mat_bytes = ((uint64_t)kMaxColsWide + 64) * dp_R(kMaxLen) * 64 * 8;
	dp.mat = (uint32_t*)malloc(mat_bytes); dp.masks = (uint16_t*)malloc((size_t)kMaxLen * (kMaxColsWide + 8) * 2);
	dp2.mat = (uint32_t*)malloc(mat_bytes); dp2.masks = (uint16_t*)malloc((size_t)kMaxLen * (kMaxColsWide + 8) * 2);
	dp.pmask_words = dp2.pmask_words = (uint32_t)pred_cells(kMaxLen, kMaxColsWide + 8);
	dp.pmask = (uint32_t*)calloc(dp.pmask_words, 4); dp.epoch = (uint32_t*)calloc(64, 4);
	dp2.pmask = (uint32_t*)calloc(dp2.pmask_words, 4); dp2.epoch = (uint32_t*)calloc(64, 4);
	const size_t rec_bytes = sizeof(ReadResult) + sizeof(AlnRes) * (size_t)(P.khits + 1);
	std::vector<uint8_t> resbuf(2 * rec_bytes);
	PairSummary summ;
	uint32_t pair_no = 0;
	for (bool last = false; !last; ) {
		HostBatch hb;
		if (inter) { fq1.next(hb, 4096, (size_t)1 << 30); finalize_interleaved(hb, opt); }
		else {
			std::unique_ptr<HostBatch> b1(new HostBatch()), b2(new HostBatch());
			fq1.next(*b1, 4096, (size_t)1 << 30);
			fq2.next(*b2, 4096, (size_t)1 << 30);
			merge_mate_batches(std::move(b1), std::move(b2), hb, opt);
		}
		last = hb.last;
		if (upto_hit && hb.upto_hit) *upto_hit = true;
		if (!hb.bad_input.empty()) { fprintf(stderr, "Error: %s\n", hb.bad_input.c_str()); return 1; }
		for (size_t pi = 0; pi + 1 < hb.reads.size(); pi += 2, pair_no++) {
			const ReadRec& r1 = hb.reads[pi]; const ReadRec& r2 = hb.reads[pi + 1];
			const size_t maxlen_ = (size_t)(kMaxLen < BT2G_MAX_READ_LEN ? kMaxLen : BT2G_MAX_READ_LEN);
			if (r1.seq.size() > maxlen_ || r2.seq.size() > maxlen_) { fprintf(stderr, "Error: read %s is longer than %d bp (unsupported)\n", r1.name.str().c_str(), (int)maxlen_); return 1; }
			ReadResult& rr1 = *(ReadResult*)resbuf.data();
			ReadResult& rr2 = *(ReadResult*)(resbuf.data() + rec_bytes);
			g_rp = hb.rp[pi]; g_Pp = &P; g_ixp = &ix;
			g_st.max_cols = (uint32_t)kMaxColsWide;      // (the product's driver asks bt2g_align_batch for what its pairs need, up to this: bt2g_align_params::max_dp_cols)
		Aligner<TOff, HostPlat> al(*w, dp);
			g_st.dp_main = dp; g_st.dp_opp = dp2;
			g_st.pe_seq[0] = (const uint8_t*)r1.seq.data(); g_st.pe_qual[0] = (const uint8_t*)r1.qual.data(); g_st.pe_len[0] = (uint32_t)r1.seq.size();
			g_st.pe_seq[1] = (const uint8_t*)r2.seq.data(); g_st.pe_qual[1] = (const uint8_t*)r2.qual.data(); g_st.pe_len[1] = (uint32_t)r2.seq.size();
			g_st.pe_rp[0] = hb.rp[pi]; g_st.pe_rp[1] = hb.rp[pi + 1];
			g_st.pe_pair = 0;
			al.run_pair(rr1, rr2);
			if (rr1.status || rr2.status) fprintf(stderr, "Warning: pair %s overflowed a fixed-capacity buffer (status %d, site %u %u)\n", r1.name.str().c_str(), rr1.status | rr2.status, rr1.pad2 & 0xffffu, rr2.pad2 & 0xffffu);
			summ.add(rr1, rr2);
			std::vector<const AlnRes*> a1, a2;
			for (uint32_t i = 0; i < rr1.nreport; i++) a1.push_back(&rr1.alns[i]);
			for (uint32_t i = 0; i < rr2.nreport; i++) a2.push_back(&rr2.alns[i]);
			o.clear();
			sam_pair_records(o, opt, ref, r1, r2, rr1, rr2, a1.data(), a2.data());
			fwrite(o.data(), 1, o.size(), out);
			if (metrics) fprintf(stderr, "MET\t%s\titers=%u dps=%u matedps=%u ugs=%u red=%u bt=%u n1=%u n2=%u type=%u\n", r1.name.str().c_str(),
			                     rr1.n_ex_iters, rr1.n_ex_dps, rr1.n_mate_dps, rr1.n_ex_ugs, rr1.n_redundants, rr1.n_bt_attempts, rr1.nalns, rr2.nalns, rr1.pair_type);
		}
	}
	if (keep) *keep = summ;
	else if (!opt.quiet) summ.print(stderr, !opt.no_discordant, !opt.no_mixed);
	return 0;
}

int main(int argc, char** argv) {
	Options opt;
	CliExtra& ex = g_ex;
	ex.allow_paired = true;
	const std::string perr = parse_cli(argc, argv, opt, ex);
	if (ex.arg_desc) { print_arg_desc(); return 0; }
	if (ex.version) { print_version(argv[0]); return 0; }
	if (ex.help) { print_usage(argv[0]); return 0; }
	if (!perr.empty()) { fprintf(stderr, "%s\n", perr.c_str()); return 1; }
	const bool metrics = ex.metrics;
	opt.cmdline = "hostsim";
	HostIndex hidx;
	std::string err;
	if (load_index(opt.index_base, hidx, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
	if (getenv("BT2G_INDEX_DUMP")) {
		// tests/test_rank_index.py: every row's view through the HBM layout (rank blocks, full suffix array), to be compared with the oracle
		auto dump = [&](auto tag) {
			using TOff = decltype(tag);
			DevIndex<TOff> ix;
			make_dev_index(hidx, ix);
			for (uint64_t r = 0; r <= (uint64_t)ix.fw.len; r++) {
				uint32_t steps = 0;
				const TOff jo = get_offset(ix.fw, (TOff)r, steps);
				TOff f4[4], b4[4];
				rank4(ix.fw, (TOff)r, f4); rank4(ix.bw, (TOff)r, b4);
				TOff rr = (TOff)r;
				const int ch = map_lf1(ix.fw, rr);
				TOff t1 = 0, b1 = 0;
				const int ns = rank1_pair(ix.fw, (TOff)r, (TOff)(r + 37 <= (uint64_t)ix.fw.len ? r + 37 : ix.fw.len), (int)(r & 3), t1, b1);
				printf("%llu %llu %u %llu %llu %llu %llu %d %llu %llu %llu %llu %llu %llu %llu %d\n", (unsigned long long)r, (unsigned long long)jo, steps,
				       (unsigned long long)f4[0], (unsigned long long)f4[1], (unsigned long long)f4[2], (unsigned long long)f4[3], ch, (unsigned long long)(ch < 0 ? 0 : rr),
				       (unsigned long long)b4[0], (unsigned long long)b4[1], (unsigned long long)b4[2], (unsigned long long)b4[3], (unsigned long long)t1, (unsigned long long)b1, ns);
			}
		};
		if (hidx.off_size == 4) dump((uint32_t)0); else dump((uint64_t)0);
		return 0;
	}
	FILE* out = opt.out_file.empty() ? stdout : fopen(opt.out_file.c_str(), "wb");
	int rc;
	if (opt.mixed_unpaired) {
		// pairs first, then the unpaired reads, one summary (the product's reader thread does the same, bt2g_search.cpp)
		PairSummary ps; AlnSummary us;
		Options po = opt; po.reads_file.clear();
		Options uo = opt; uo.paired = false; uo.mate1_file.clear(); uo.mate2_file.clear(); uo.interleaved_file.clear();
		bool upto_hit = false;      // -u ended the pairs: the run ends there (the reference's composer reports "done")
		rc = hidx.off_size == 4 ? run_pairs<uint32_t>(hidx, po, out, metrics, &ps, &upto_hit) : run_pairs<uint64_t>(hidx, po, out, metrics, &ps, &upto_hit);
		if (rc == 0 && !upto_hit) rc = hidx.off_size == 4 ? run<uint32_t>(hidx, uo, out, metrics, false, &us) : run<uint64_t>(hidx, uo, out, metrics, false, &us);
		if (rc == 0 && !opt.quiet) print_mixed_summary(stderr, ps, us, !opt.no_discordant, !opt.no_mixed);
	} else
	if (opt.paired) rc = hidx.off_size == 4 ? run_pairs<uint32_t>(hidx, opt, out, metrics) : run_pairs<uint64_t>(hidx, opt, out, metrics);
	else rc = hidx.off_size == 4 ? run<uint32_t>(hidx, opt, out, metrics) : run<uint64_t>(hidx, opt, out, metrics);
	if (out != stdout) fclose(out);
	return rc;
}
