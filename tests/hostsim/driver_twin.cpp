// tests/hostsim/driver_twin.cpp -- TEST-ONLY: the product's aligner driver (bowtie2_amd/csrc/bt2g_search.cpp, the code behind
// extern "C" bowtie()) compiled unchanged against (a) a dozen HIP runtime calls on host memory (fakehip/) and (b) the C ABI of
// include/bt2g.h implemented with the host-compiled worker of hostsim.cpp.  What this checks without a GPU is the driver itself: the
// reader / device-stage / writer threads, ordered output with several workers, small --batch values, --shard blocks and the index file
// they write, the switch from pair batches to unpaired batches of a mixed run, packed result records.  The alignments come from the same
// worker source as tests/hostsim/hostsim (which has its own, simpler main()).  Never linked into libbt2g.so or the drop-in binaries.
#define main hostsim_main_unused
#include "hostsim.cpp"
#undef main

#include <mutex>

struct bt2g_ctx {
	HostIndex hidx;
	DevIndex<uint32_t> ix32;
	DevIndex<uint64_t> ix64;
	Work* w = nullptr;
	DpScratch dp, dp2;
	std::string err;
	bool loaded = false;
};
static std::mutex g_twin_mu;      // the host worker keeps its per-read state in globals: one batch at a time

extern "C" {

int bt2g_ctx_create(int, bt2g_ctx** out) {
	bt2g_ctx* c = new bt2g_ctx();
	c->w = (Work*)calloc(1, sizeof(Work));      // (zero pages on first touch)
	const uint64_t mat_bytes = ((uint64_t)kMaxColsWide + 64) * dp_R(kMaxLen) * 64 * 8;
	for (DpScratch* d : {&c->dp, &c->dp2}) {
		d->mat = (uint32_t*)malloc(mat_bytes); d->masks = (uint16_t*)malloc((size_t)kMaxLen * (kMaxColsWide + 8) * 2);
		d->pmask_words = (uint32_t)pred_cells(kMaxLen, kMaxColsWide + 8);
		d->pmask = (uint32_t*)calloc(d->pmask_words, 4); d->epoch = (uint32_t*)calloc(64, 4);
	}
	*out = c;
	return 0;
}
void bt2g_ctx_destroy(bt2g_ctx* c) { delete c; }
const char* bt2g_last_error(const bt2g_ctx* c) { return c ? c->err.c_str() : "no context"; }
int bt2g_index_load(bt2g_ctx* c, const char* base) {
	if (load_index(base, c->hidx, c->err)) return 1;
	if (c->hidx.off_size == 4) make_dev_index(c->hidx, c->ix32); else make_dev_index(c->hidx, c->ix64);
	c->loaded = true;
	return 0;
}
int bt2g_index_info_get(const bt2g_ctx* c, bt2g_index_info* out) {
	memset(out, 0, sizeof(*out));
	out->n_pat = c->hidx.fw.n_pat; out->off_size = (uint32_t)c->hidx.off_size;
	return 0;
}
int bt2g_index_refname(const bt2g_ctx* c, uint64_t tidx, const char** name, uint64_t* len) {
	*name = c->hidx.fw.refnames[tidx].c_str(); *len = c->hidx.plen_at(tidx);
	return 0;
}
uint64_t bt2g_align_result_stride(uint32_t khits) {       // as bt2g_capi.hip
	if (khits == 0) khits = 1;
	const uint64_t slot = sizeof(AlnRes);      // (the twin's worker is compiled with the largest capacities of every class: its slots are the long-read class's, for any -k)
	const uint64_t b = offsetof(bt2g_read_result, alns) + (uint64_t)khits * slot;
	return (b + 15) & ~(uint64_t)15;
}

} // extern "C"

template <typename TOff>
static void twin_align(bt2g_ctx* c, const DevIndex<TOff>& ix, const bt2g_reads* reads, const ReadParams* rp, const AlignParams& P, uint8_t* results, uint64_t stride) {
	const size_t rec_bytes = sizeof(ReadResult) + sizeof(AlnRes) * (size_t)(P.khits + 1);
	std::vector<uint8_t> resbuf(2 * rec_bytes);
	const uint32_t n = reads->n_reads;
	auto seq = [&](uint32_t i) { return reads->d_seq + reads->d_off[i]; };
	auto qual = [&](uint32_t i) { return reads->d_qual + reads->d_off[i]; };
	auto len = [&](uint32_t i) { return (uint32_t)(reads->d_off[i + 1] - reads->d_off[i]); };
	if (P.paired) {
		for (uint32_t i = 0; i + 1 < n; i += 2) {
			ReadResult& rr1 = *(ReadResult*)resbuf.data();
			ReadResult& rr2 = *(ReadResult*)(resbuf.data() + rec_bytes);
			g_rp = rp[i]; g_Pp = &P; g_ixp = &ix;
			g_st.max_cols = dp_cols_for(P);
			Aligner<TOff, HostPlat> al(*c->w, c->dp);
			g_st.dp_main = c->dp; g_st.dp_opp = c->dp2;
			for (int m = 0; m < 2; m++) { g_st.pe_seq[m] = seq(i + m); g_st.pe_qual[m] = qual(i + m); g_st.pe_len[m] = len(i + m); g_st.pe_rp[m] = rp[i + m]; }
			g_st.pe_pair = 0;
			al.run_pair(rr1, rr2);
			memcpy(results + (uint64_t)i * stride, &rr1, std::min<uint64_t>(stride, rec_bytes));
			memcpy(results + (uint64_t)(i + 1) * stride, &rr2, std::min<uint64_t>(stride, rec_bytes));
		}
		return;
	}
	for (uint32_t i = 0; i < n; i++) {
		ReadResult& rr = *(ReadResult*)resbuf.data();
		g_rp = rp[i]; g_Pp = &P; g_ixp = &ix;
		g_hot.len = len(i);
		memcpy(g_hot.seq, seq(i), g_hot.len);
		memcpy(g_hot.qual, qual(i), g_hot.len);
		g_st.max_cols = dp_cols_for(P);
		Aligner<TOff, HostPlat> al(*c->w, c->dp);
		al.run(rr);
		memcpy(results + (uint64_t)i * stride, &rr, std::min<uint64_t>(stride, rec_bytes));
	}
}

extern "C" {

int bt2g_align_batch(bt2g_ctx* c, const bt2g_reads* reads, const bt2g_read_params* d_rparams, const bt2g_align_params* params, uint32_t, void* d_results, void*) {
	// the product's C ABI keeps a working set per stream and may be called from the driver's device-stage threads at once; the host
	// worker behind this twin has ONE (static) working set, so the calls take turns here
	static std::mutex one_at_a_time;
	std::lock_guard<std::mutex> guard(one_at_a_time);
	if (!c->loaded) { c->err = "no index loaded"; return 1; }
	if (params->paired && (reads->n_reads & 1u)) { c->err = "paired mode needs an even number of reads (mates interleaved)"; return 1; }
	const AlignParams& P = *(const AlignParams*)params;
	if (getenv("BT2G_TWIN_NULL_ALIGNER")) {
		// host-pipeline profiling aid (tools/host_pipeline_rate.sh): no alignment at all, every read "aligns" end to end without edits at a
		// position made up from its index, so that parsing, packing, SAM formatting and writing can be timed at their own speed
		const uint64_t stride = bt2g_align_result_stride((uint32_t)params->khits);
		bt2g_index_info inf; bt2g_index_info_get(c, &inf);
		for (uint32_t i = 0; i < reads->n_reads; i++) {
			ReadResult* rr = (ReadResult*)((uint8_t*)d_results + (uint64_t)i * stride);
			memset(rr, 0, offsetof(ReadResult, alns) + sizeof(AlnRes));
			const uint32_t len = (uint32_t)(reads->d_off[i + 1] - reads->d_off[i]);
			rr->aligned = 1; rr->nalns = 1; rr->nreport = 1; rr->filt = 15; rr->best = 0; rr->secbest = 0;
			AlnRes& a = rr->alns[0];
			a.refid = 0; a.refoff = (int64_t)((uint64_t)i * 97 % (c->hidx.plen_at(0) > len ? c->hidx.plen_at(0) - len : 1)); a.reflen = (int64_t)c->hidx.plen_at(0); a.rdlen = (uint16_t)len; a.rfextent = (uint16_t)len; a.rdextent = (uint16_t)len; a.fw = (i & 1); a.score = 0; a.nned = 0;
		}
		return 0;
	}
	std::lock_guard<std::mutex> g(g_twin_mu);
	const uint64_t stride = bt2g_align_result_stride((uint32_t)params->khits);
	if (c->hidx.off_size == 4) twin_align<uint32_t>(c, c->ix32, reads, (const ReadParams*)d_rparams, P, (uint8_t*)d_results, stride);
	else twin_align<uint64_t>(c, c->ix64, reads, (const ReadParams*)d_rparams, P, (uint8_t*)d_results, stride);
	return 0;
}

// packed records as k_pack_sizes / k_pack_scan / k_pack_copy lay them out (bt2g_kernels.hip): header, then the reported alignments
// cut after their last edit, 8-byte aligned; offs[i] = start of record i, offs[n] = total
int bt2g_results_pack(bt2g_ctx*, const void* d_results, uint32_t n, uint32_t khits, void* d_packed, uint64_t* offs, void*) {
	if (khits == 0) khits = 1;
	const uint64_t stride = bt2g_align_result_stride(khits);
	const uint32_t head = (uint32_t)offsetof(bt2g_read_result, alns), ahead = (uint32_t)offsetof(bt2g_aln, ned);
	uint64_t pos = 0;
	for (uint32_t r = 0; r < n; r++) {
		offs[r] = pos;
		const uint8_t* src = (const uint8_t*)d_results + (uint64_t)r * stride;
		const bt2g_read_result* rr = (const bt2g_read_result*)src;
		uint8_t* dst = (uint8_t*)d_packed + pos;
		const uint32_t na = rr->aligned ? std::min<uint32_t>(rr->nreport, khits) : 0u;
		memcpy(dst, src, head);
		((bt2g_read_result*)dst)->nreport = na;
		pos += head;
		const uint32_t slot = (rr->pad2 >> 16) ? (rr->pad2 >> 16) * 8u : (uint32_t)sizeof(bt2g_aln);      // the record says how large its alignment slots are
		for (uint32_t k = 0; k < na; k++) {
			const bt2g_aln* a = (const bt2g_aln*)((const uint8_t*)rr->alns + (uint64_t)k * slot);
			const uint32_t bytes = (ahead + std::min<uint32_t>(a->nned, (slot - ahead) / (uint32_t)sizeof(bt2g_edit)) * (uint32_t)sizeof(bt2g_edit) + 7u) & ~7u;
			memcpy((uint8_t*)d_packed + pos, a, bytes);
			pos += bytes;
		}
	}
	offs[n] = pos;
	return 0;
}

} // extern "C"

#include "../../bowtie2_amd/csrc/bt2g_search.cpp"

int main(int argc, char** argv) { return bowtie(argc, const_cast<const char**>(argv)); }
