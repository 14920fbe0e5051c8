// bt2g_capi.hip -- implementation of the C ABI declared in include/bt2g.h.
//
// Owns the HBM-resident index and the per-context device scratch; every compute entry point
// is a thin, asynchronous launch of the kernels in bt2g_kernels.hip.  There is deliberately
// no host fallback: without a usable gfx950 device every entry point returns
// BT2G_ERR_NO_DEVICE.
#include "../../include/bt2g.h"
#include "bt2g_index.hpp"
#include "bt2g_kernels.hpp"
#include "bt2g_align_kernel.hpp"
#include "bt2g_rankidx.hpp"

#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace bt2g;

struct bt2g_ctx {
	int device = -1;
	std::string err;
	bool loaded = false;
	int off_size = 0;
	HostIndex* host = nullptr;           // header fields + names (bulk arrays are dropped after upload)
	DevIndex<uint32_t> ix32;
	DevIndex<uint64_t> ix64;
	std::vector<void*> allocs;           // index allocations in HBM
	uint64_t hbm_bytes = 0;
	DevCounters* d_cnt = nullptr;
	uint8_t* d_dp_scratch = nullptr;     // wavefront-layout DP scratch
	uint64_t dp_scratch_bytes = 0;
	uint32_t n_cu = 0;
	// What one bt2g_align_batch call works in.  A context keeps one such set PER STREAM it has been called on (up to kMaxSlots), so that
	// batches issued on different streams -- the product driver's device-stage threads, bench.py's two alternating streams -- run side by
	// side: the copies of one overlap the kernels of another, and the tail of one batch's worker kernel (a few pathological reads can keep
	// a handful of waves busy long after the rest of the batch is done) is filled by the next batch's waves.
	struct BatchSlot {
		hipStream_t owner = nullptr;
		bool used = false;
		uint8_t* d_arena = nullptr;          // per-wave Work + DP scratch of the fused worker
		uint64_t arena_bytes = 0;
		uint64_t arena_layout = 0;           // fingerprint of the arena's carving (epoch-tagged masks are only valid within one)
		unsigned int* d_next = nullptr;      // work-queue head, small counters, the worker's phase profile
		uint8_t* d_pre = nullptr;            // batch pre-computation (sweep, round-0 seed hits, extensions, 1-mm hits)
		uint64_t pre_bytes = 0;
		hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // kernel boundaries of the last align batch
		bool ev_valid = false;
		int busy = 0;                        // bt2g_align_batch calls inside this slot right now (under slot_mu): a busy slot never changes owner
		bool fence_pending = false;          // a batch failed after kernels were queued: the owner stream is synchronised before the slot changes owner
		uint64_t last_use = 0;               // batch number of the slot's last bt2g_align_batch: the least recently used slot is handed to a new stream
		uint64_t arena_cut_stride = 0;       // != 0: the arena holds fewer waves than a full launch of this per-wave stride wants (it was cut to the memory budget)
	};
	static constexpr int kMaxSlots = 4;
	BatchSlot slots[kMaxSlots];
	std::mutex slot_mu;
	int last_slot = -1;                  // slot of the most recent bt2g_align_batch (bt2g_align_timing_read); read and written under slot_mu
	uint64_t n_batches = 0;
	bool precomp = true;                 // BT2G_NO_PRECOMP=1: the worker computes every FM phase itself (A/B testing)
};

namespace {

int fail(bt2g_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

int hip_fail(bt2g_ctx* c, hipError_t e, const char* what) {
	return fail(c, BT2G_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

template <typename T>
int upload(bt2g_ctx* c, const void* src, uint64_t nbytes, const T** dst) {
	void* p = nullptr;
	const uint64_t alloc = nbytes ? ((nbytes + 255) & ~255ull) : 256;
	hipError_t e = hipMalloc(&p, alloc);
	if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(index)");
	c->allocs.push_back(p);
	c->hbm_bytes += alloc;
	if (nbytes) {
		e = hipMemcpy(p, src, nbytes, hipMemcpyHostToDevice);
		if (e != hipSuccess) return hip_fail(c, e, "hipMemcpy(index)");
	}
	*dst = reinterpret_cast<const T*>(p);
	return 0;
}

// Streams sections of the index files into device memory.  kThreads reader threads pread() chunks into pinned buffers (two per thread) and
// hand each chunk to the copy engine on the thread's own stream: the page cache is read by several cores at once, no second host copy of the
// 4-5 GB of a genome-sized index is made, and the copies overlap the transcoding kernels of the sections already on the device.
class FileStreamer {
public:
	static constexpr size_t kChunk = 8u << 20;
	static constexpr int kThreads = 6, kBufs = 2;
	explicit FileStreamer(int device) : device_(device) {}
	~FileStreamer() {
		for (void* p : pinned_) if (p) (void)hipHostFree(p);
		for (hipEvent_t e : ev_) if (e) (void)hipEventDestroy(e);
		for (hipStream_t st : st_) if (st) (void)hipStreamDestroy(st);
	}
	bool init() {
		for (int t = 0; t < kThreads; t++) {
			hipStream_t st = nullptr;
			if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
			st_.push_back(st);
			for (int b = 0; b < kBufs; b++) {
				void* p = nullptr; hipEvent_t e = nullptr;
				if (hipHostMalloc(&p, kChunk, hipHostMallocDefault) != hipSuccess) return false;
				pinned_.push_back(p);
				if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
				ev_.push_back(e);
			}
		}
		return true;
	}
	// file section -> dst (device); returns when every byte has arrived.  0, or 1 = read error, 2 = HIP error
	int put(const FileSpan& sp, void* dst) {
		if (sp.nbytes == 0) return 0;
		const int fd = open(sp.path.c_str(), O_RDONLY | O_CLOEXEC);
		if (fd < 0) return 1;
		(void)posix_fadvise(fd, (off_t)sp.off, (off_t)sp.nbytes, POSIX_FADV_SEQUENTIAL);
		const uint64_t nch = (sp.nbytes + kChunk - 1) / kChunk;
		std::atomic<uint64_t> next{0};
		std::atomic<int> bad{0};
		auto work = [&](int t) {
			if (hipSetDevice(device_) != hipSuccess) { bad = 2; return; }
			bool used[kBufs] = {};
			int b = 0;
			for (;;) {
				const uint64_t i = next.fetch_add(1);
				if (i >= nch || bad.load()) break;
				const uint64_t at = i * (uint64_t)kChunk;
				const size_t n = (size_t)(sp.nbytes - at < (uint64_t)kChunk ? sp.nbytes - at : (uint64_t)kChunk);
				uint8_t* buf = (uint8_t*)pinned_[t * kBufs + b];
				if (used[b] && hipEventSynchronize(ev_[t * kBufs + b]) != hipSuccess) { bad = 2; break; }      // the copy that last used this buffer
				size_t got = 0;
				while (got < n) {
					const ssize_t r = pread(fd, buf + got, n - got, (off_t)(sp.off + at + got));
					if (r < 0 && errno == EINTR) continue;
					if (r <= 0) { bad = 1; break; }
					got += (size_t)r;
				}
				if (bad.load()) break;
				if (hipMemcpyAsync((uint8_t*)dst + at, buf, n, hipMemcpyHostToDevice, st_[t]) != hipSuccess ||
				    hipEventRecord(ev_[t * kBufs + b], st_[t]) != hipSuccess) { bad = 2; break; }
				used[b] = true; b = (b + 1) % kBufs;
			}
			if (hipStreamSynchronize(st_[t]) != hipSuccess) bad = 2;
		};
		const int nt = (int)(nch < (uint64_t)kThreads ? nch : (uint64_t)kThreads);
		std::vector<std::thread> th;
		// (a C-ABI entry point: std::thread may throw std::system_error under a thread limit -- the chunks are then read by the threads there are)
		try { th.reserve((size_t)nt); for (int t = 1; t < nt; t++) th.emplace_back(work, t); } catch (...) {}
		work(0);
		for (auto& x : th) x.join();
		close(fd);
		return bad.load();
	}
private:
	int device_;
	std::vector<hipStream_t> st_;
	std::vector<void*> pinned_;
	std::vector<hipEvent_t> ev_;
};

// What one bt2g_index_load works with besides the context: the file streamer, the stream the transcoding kernels run on (non-blocking: the
// copies of the next section overlap them), the temporary device copies of file sections (freed at the end, after one synchronisation), and
// a stopwatch (BT2G_DEBUG_LOAD=1 prints the phases).
struct LoadJob {
	bt2g_ctx* c;
	FileStreamer fs;
	hipStream_t cs = nullptr;
	std::vector<void*> temps;
	unsigned long long* d_lost = nullptr;
	int off_rate = 0;
	bool verbose;
	std::chrono::steady_clock::time_point t0;
	explicit LoadJob(bt2g_ctx* ctx) : c(ctx), fs(ctx->device), verbose(getenv("BT2G_DEBUG_LOAD") != nullptr), t0(std::chrono::steady_clock::now()) {}
	~LoadJob() {
		if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
		for (void* p : temps) (void)hipFree(p);
	}
	void lap(const char* what) {
		if (!verbose) return;
		const auto t1 = std::chrono::steady_clock::now();
		fprintf(stderr, "[bt2g] index load: %-34s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
		t0 = t1;
	}
	// a temporary device copy of a section: from its file (lazy load) or from the host vector
	int put(const FileSpan& sp, const std::vector<uint8_t>& v, uint64_t pad, void** out) {
		const uint64_t nbytes = sp.path.empty() ? v.size() : sp.nbytes;
		void* p = nullptr;
		hipError_t e = hipMalloc(&p, nbytes + pad ? nbytes + pad : 256);
		if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(index section)");
		temps.push_back(p);
		*out = p;
		return fill(sp, v, p);
	}
	int fill(const FileSpan& sp, const std::vector<uint8_t>& v, void* p) {
		if (sp.path.empty()) {
			hipError_t e = v.empty() ? hipSuccess : hipMemcpy(p, v.data(), v.size(), hipMemcpyHostToDevice);
			return e == hipSuccess ? 0 : hip_fail(c, e, "hipMemcpy(index section)");
		}
		const int r = fs.put(sp, p);
		if (r == 1) return fail(c, BT2G_ERR_IO, "cannot read " + sp.path);
		if (r) return fail(c, BT2G_ERR_HIP, "copying " + sp.path + " to the device failed");
		return 0;
	}
};

template <typename T>
int alloc_index(bt2g_ctx* c, uint64_t nbytes, T** dst) {
	void* p = nullptr;
	const uint64_t alloc = nbytes ? ((nbytes + 255) & ~255ull) : 256;
	hipError_t e = hipMalloc(&p, alloc);
	if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(index)");
	c->allocs.push_back(p);
	c->hbm_bytes += alloc;
	*dst = reinterpret_cast<T*>(p);
	return 0;
}

// One direction of the index.  The BWT sides and (forward index) the SA sample go to the device as they are in the files and are
// transcoded there into the layout the kernels query: rank blocks and the full suffix array (bt2g_device.hpp, bt2g_rankidx.hpp).
// The kernels are queued on the job's stream and not waited for: the caller streams the next section meanwhile (upload_index).
template <typename TOff>
int upload_ebwt(LoadJob& L, const HostEbwt& h, bool fw, DevEbwt<TOff>& d) {
	bt2g_ctx* c = L.c;
	int rc;
	if ((rc = upload(c, h.ftab.data(), h.ftab.size(), &d.ftab))) return rc;
	if ((rc = upload(c, h.eftab.data(), h.eftab.size(), &d.eftab))) return rc;
	d.sa = nullptr;
	d.len = (TOff)h.len; d.zoff = (TOff)h.zoff;
	d.zblk = (uint64_t)h.zoff >> kBlkShift; d.zchar = (uint32_t)((uint64_t)h.zoff & (kBlkLen - 1));
	for (int i = 0; i < 5; i++) d.fchr[i] = (TOff)h.fchr[i];
	d.ftab_chars = (uint32_t)h.ftab_chars; d.off_rate = (uint32_t)h.off_rate; d.is_fw = fw ? 1 : 0;
	{
		void* sides = nullptr;
		const uint64_t n_sides = h.ebwt_tot_len / OffTraits<TOff>::kSideSz;
		const uint64_t n_blocks = rank_block_count(n_sides, OffTraits<TOff>::kSideBwtLen);
		RankBlock* blk = nullptr;
		if ((rc = alloc_index(c, n_blocks * sizeof(RankBlock), &blk))) return rc;
		if ((rc = L.put(h.ebwt_span, h.ebwt, 0, &sides))) return rc;
		L.lap(fw ? "BWT sides -> device" : "mirror BWT sides -> device");
		hipError_t e = launch_make_rank_blocks<TOff>((const uint8_t*)sides, n_sides, d.fchr, d.zoff, blk, n_blocks, L.cs);
		if (e != hipSuccess) return hip_fail(c, e, "k_make_rank_blocks");
		d.blk = blk;
	}
	if (fw) {
		void* offs = nullptr;
		uint64_t* sa = nullptr;
		if ((rc = alloc_index(c, ((uint64_t)h.len + 1) * sizeof(uint64_t), &sa))) return rc;
		void* lost = nullptr;
		hipError_t e = hipMalloc(&lost, 256);
		if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(lost rows)");
		L.temps.push_back(lost);
		if ((e = hipMemsetAsync(lost, 0, 256, L.cs)) != hipSuccess) return hip_fail(c, e, "hipMemsetAsync(lost rows)");
		if ((rc = L.put(h.offs_span, h.offs, 0, &offs))) return rc;
		L.lap("SA sample -> device");
		e = launch_make_full_sa<TOff>(d, (const TOff*)offs, sa, (unsigned long long*)lost, L.cs);
		if (e != hipSuccess) return hip_fail(c, e, "k_sa_segments");
		L.d_lost = (unsigned long long*)lost; L.off_rate = (int)h.off_rate;
		d.sa = sa;
	}
	return 0;
}

constexpr int kLoadRetryEager = 1;      // upload_index: the streamed load could not start (no bt2g_status is positive)

template <typename TOff>
int upload_index(bt2g_ctx* c, const HostIndex& h, DevIndex<TOff>& d) {
	LoadJob L(c);
	if (hipStreamCreateWithFlags(&L.cs, hipStreamNonBlocking) != hipSuccess) return fail(c, BT2G_ERR_HIP, "index load: cannot create a stream");
	// (pinned buffers only when sections are to be streamed; a host that will not pin 96 MB gets the load through host memory instead)
	if (!h.fw.ebwt_span.path.empty() && !L.fs.init()) return kLoadRetryEager;
	L.lap("streams + pinned buffers");
	int rc;
	if ((rc = upload_ebwt(L, h.fw, true, d.fw))) return rc;
	if ((rc = upload_ebwt(L, h.bw, false, d.bw))) return rc;      // (streams in while the forward index's suffix array is being made)
	if ((rc = upload(c, h.fw.rstarts.data(), h.fw.rstarts.size(), &d.rstarts))) return rc;
	if ((rc = upload(c, h.fw.plen.data(), h.fw.plen.size(), &d.plen))) return rc;
	d.n_frag = (TOff)h.fw.n_frag; d.n_pat = (TOff)h.fw.n_pat;
	const HostRef& r = h.ref;
	if ((rc = upload(c, r.rec_refpos.data(), r.rec_refpos.size() * 8, &d.ref.rec_refpos))) return rc;
	if ((rc = upload(c, r.rec_bufpos.data(), r.rec_bufpos.size() * 8, &d.ref.rec_bufpos))) return rc;
	if ((rc = upload(c, r.rec_len.data(), r.rec_len.size() * 8, &d.ref.rec_len))) return rc;
	if ((rc = upload(c, r.ref_rec_offs.data(), r.ref_rec_offs.size() * 8, &d.ref.ref_rec_offs))) return rc;
	if ((rc = upload(c, r.ref_lens.data(), r.ref_lens.size() * 8, &d.ref.ref_lens))) return rc;
	if (r.buf_span.path.empty()) { if ((rc = upload(c, r.buf.data(), r.buf.size(), &d.ref.buf))) return rc; }
	else {
		// (16 bytes of slack behind the 2-bit reference: device code may read a few bytes past the end)
		uint8_t* buf = nullptr;
		if ((rc = alloc_index(c, r.buf_span.nbytes + 16, &buf))) return rc;
		hipError_t e = hipMemsetAsync(buf + r.buf_span.nbytes, 0, 16, L.cs);
		if (e != hipSuccess) return hip_fail(c, e, "hipMemsetAsync(reference slack)");
		if ((rc = L.fill(r.buf_span, r.buf, buf))) return rc;
		d.ref.buf = buf;
	}
	d.ref.nrefs = r.nrefs;
	L.lap("2-bit reference + tables -> device");
	hipError_t e = hipStreamSynchronize(L.cs);
	if (e != hipSuccess) return hip_fail(c, e, "index transcoding kernels");
	L.lap("transcoding kernels (what is left)");
	if (L.d_lost) {
		unsigned long long n_lost = 0;
		e = hipMemcpy(&n_lost, L.d_lost, sizeof n_lost, hipMemcpyDeviceToHost);
		if (e != hipSuccess) return hip_fail(c, e, "hipMemcpy(lost rows)");
		if (n_lost) {
			char msg[256];
			snprintf(msg, sizeof msg, "the suffix-array sample of this index (--offrate %d) leaves %llu rows more than 65534 LF steps from a sampled row; "
			         "this build keeps the step count in 16 bits and will not load it (rebuild with a smaller --offrate)", L.off_rate, n_lost);
			return fail(c, BT2G_ERR_UNSUPPORTED, msg);
		}
	}
	return 0;
}

void free_index(bt2g_ctx* c) {
	for (void* p : c->allocs) (void)hipFree(p);
	c->allocs.clear();
	c->hbm_bytes = 0;
	c->loaded = false;
	delete c->host; c->host = nullptr;
}

int need_loaded(bt2g_ctx* c) {
	if (!c) return BT2G_ERR_ARG;
	if (!c->loaded) return fail(c, BT2G_ERR_ARG, "no index loaded");
	hipError_t e = hipSetDevice(c->device);
	if (e != hipSuccess) return hip_fail(c, e, "hipSetDevice");
	return 0;
}

} // namespace

extern "C" {

uint32_t bt2g_version(void) { return (0u << 16) | 1u; }

int bt2g_ctx_create(int device, bt2g_ctx** out) {
	if (!out) return BT2G_ERR_ARG;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BT2G_ERR_NO_DEVICE;
	if (hipSetDevice(device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	// the fat binary only carries gfx950 code objects
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BT2G_ERR_NO_DEVICE;
	bt2g_ctx* c = new (std::nothrow) bt2g_ctx();
	if (!c) return BT2G_ERR_NOMEM;
	c->device = device;
	c->n_cu = (uint32_t)prop.multiProcessorCount;
	{ const char* v = std::getenv("BT2G_NO_PRECOMP"); c->precomp = !(v && v[0] == '1'); }
	if (hipMalloc((void**)&c->d_cnt, kCntSlots * sizeof(DevCounters)) != hipSuccess) { delete c; return BT2G_ERR_HIP; }
	(void)hipMemset(c->d_cnt, 0, kCntSlots * sizeof(DevCounters));
	*out = c;
	return 0;
}

void bt2g_ctx_destroy(bt2g_ctx* c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	free_index(c);
	if (c->d_cnt) (void)hipFree(c->d_cnt);
	if (c->d_dp_scratch) (void)hipFree(c->d_dp_scratch);
	for (auto& S : c->slots) {
		if (S.d_arena) (void)hipFree(S.d_arena);
		if (S.d_pre) (void)hipFree(S.d_pre);
		for (int i = 0; i < 7; i++) if (S.ev[i]) (void)hipEventDestroy(S.ev[i]);
		if (S.d_next) (void)hipFree(S.d_next);
	}
	delete c;
}

const char* bt2g_last_error(const bt2g_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int bt2g_index_load(bt2g_ctx* c, const char* base) {
	if (!c || !base) return BT2G_ERR_ARG;
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	free_index(c);
	c->host = new HostIndex();
	std::string err;
	// (the large sections stay in their files and are streamed to the device; BT2G_LOAD_SERIAL=1: through host memory, as before round 5)
	static const bool serial = getenv("BT2G_LOAD_SERIAL") != nullptr;
	int rc = 0;
	for (int lazy = serial ? 0 : 1; lazy >= 0; lazy--) {
		rc = load_index(base, *c->host, err, lazy != 0);
		if (rc) { fail(c, rc, err); delete c->host; c->host = nullptr; return rc; }
		c->off_size = c->host->off_size;
		rc = (c->off_size == 4) ? upload_index(c, *c->host, c->ix32) : upload_index(c, *c->host, c->ix64);
		if (rc != kLoadRetryEager) break;
		free_index(c);
		c->host = new HostIndex();
	}
	if (rc) { free_index(c); return rc; }
	// keep only header fields, names and lengths on the host
	HostIndex& h = *c->host;
	for (HostEbwt* e : {&h.fw, &h.bw}) {
		std::vector<uint8_t>().swap(e->ebwt); std::vector<uint8_t>().swap(e->ftab);
		std::vector<uint8_t>().swap(e->offs); std::vector<uint8_t>().swap(e->rstarts);
	}
	std::vector<uint8_t>().swap(h.ref.buf);
	c->loaded = true;
	return 0;
}

int bt2g_index_info_get(const bt2g_ctx* c, bt2g_index_info* o) {
	if (!c || !o || !c->loaded) return BT2G_ERR_ARG;
	const HostIndex& h = *c->host;
	o->off_size = h.off_size; o->line_rate = h.fw.line_rate; o->off_rate = h.fw.off_rate; o->ftab_chars = h.fw.ftab_chars;
	o->len = h.fw.len; o->n_pat = h.fw.n_pat; o->n_frag = h.fw.n_frag;
	o->zoff_fw = h.fw.zoff; o->zoff_bw = h.bw.zoff;
	o->ebwt_bytes = h.fw.ebwt_tot_len; o->offs_len = h.fw.offs_len; o->hbm_bytes = c->hbm_bytes;
	o->side_sz = h.fw.side_sz;
	return 0;
}

int bt2g_index_refname(const bt2g_ctx* c, uint64_t tidx, const char** name, uint64_t* len) {
	if (!c || !c->loaded || tidx >= c->host->fw.n_pat) return BT2G_ERR_ARG;
	if (name) *name = tidx < c->host->fw.refnames.size() ? c->host->fw.refnames[tidx].c_str() : "";
	if (len) *len = c->host->plen_at(tidx);
	return 0;
}

int bt2g_exact_sweep(bt2g_ctx* c, const bt2g_reads* reads, int nofw, int norc, uint32_t mine_max,
                     bt2g_sweep_out* d_out, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	if (!reads || !d_out || mine_max == 0) return fail(c, BT2G_ERR_ARG, "bad argument");
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = (c->off_size == 4)
		? launch_exact_sweep(c->ix32, *reads, nofw, norc, mine_max, d_out, c->d_cnt, st)
		: launch_exact_sweep(c->ix64, *reads, nofw, norc, mine_max, d_out, c->d_cnt, st);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_exact_sweep");
}

int bt2g_seed_search_exact(bt2g_ctx* c, const bt2g_reads* reads, const uint32_t* d_seedlen, const uint32_t* d_interval,
                           const uint32_t* d_offset, uint32_t max_seeds, bt2g_seed_hit* d_out, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	if (!reads || !d_out || !d_seedlen || !d_interval || !d_offset || max_seeds == 0) return fail(c, BT2G_ERR_ARG, "bad argument");
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = (c->off_size == 4)
		? launch_seed_search_exact(c->ix32, *reads, d_seedlen, d_interval, d_offset, nullptr, max_seeds, d_out, c->d_cnt, st)
		: launch_seed_search_exact(c->ix64, *reads, d_seedlen, d_interval, d_offset, nullptr, max_seeds, d_out, c->d_cnt, st);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_seed_search_exact");
}

// the worker's second occupancy class (bt2g_align_kernel.hip compiled again with 96 registers: 5 waves per SIMD)
extern "C" hipError_t bt2g_w5_launch_align(int off_size, const void* ix, const bt2g_align_params* P, const bt2g_reads* rd, const bt2g_read_params* d_rparams,
                                           uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride,
                                           uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof,
                                           const void* pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st);
extern "C" uint32_t bt2g_w5_waves_per_cu(void);
extern "C" uint32_t bt2g_w5_static_lds(void);
extern "C" uint64_t bt2g_w5_work_bytes(void);
extern "C" uint32_t bt2g_w5_max_len(void);
extern "C" uint32_t bt2g_w5_max_offs(void);
extern "C" void bt2g_w5_scratch_sizes(uint32_t max_len, int paired, uint32_t maxhalf, uint32_t max_cols, uint64_t* mat_bytes, uint64_t* mask_bytes, uint64_t* pmask_bytes, uint64_t* arena_stride);
// ... and its many-alignments class (-k above 64, -a: bt2g_align_kernel.hip compiled with BT2G_CLASS_BIG_K, its own Work and arena)
extern "C" hipError_t bt2g_bk_launch_align(int off_size, const void* ix, const bt2g_align_params* P, const bt2g_reads* rd, const bt2g_read_params* d_rparams,
                                           uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride,
                                           uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof,
                                           const void* pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st);
extern "C" uint32_t bt2g_bk_waves_per_cu(void);
extern "C" uint32_t bt2g_bk_max_alns(void);
extern "C" void bt2g_bk_scratch_sizes(uint32_t max_len, int paired, uint32_t maxhalf, uint32_t max_cols, uint64_t* mat_bytes, uint64_t* mask_bytes, uint64_t* pmask_bytes, uint64_t* arena_stride);
// ... and its long-read class (reads of 513 ... 1 999 bp: bt2g_align_kernel.hip compiled for 2 048 rows, 128 seed positions per strand, 2 waves per SIMD)
extern "C" hipError_t bt2g_lr_launch_align(int off_size, const void* ix, const bt2g_align_params* P, const bt2g_reads* rd, const bt2g_read_params* d_rparams,
                                           uint8_t* d_results, uint64_t result_stride, uint8_t* d_arena, uint64_t arena_stride,
                                           uint64_t mat_bytes, uint64_t mask_bytes, uint64_t pmask_bytes, uint32_t n_waves, unsigned int* d_next, unsigned long long* d_prof,
                                           const void* pre, uint32_t max_read_len, uint32_t max_cols, uint32_t lds_per_wave, hipStream_t st);
extern "C" uint32_t bt2g_lr_waves_per_cu(void);
extern "C" uint32_t bt2g_lr_max_len(void);
extern "C" uint32_t bt2g_lr_max_offs(void);
extern "C" void bt2g_lr_scratch_sizes(uint32_t max_len, int paired, uint32_t maxhalf, uint32_t max_cols, uint64_t* mat_bytes, uint64_t* mask_bytes, uint64_t* pmask_bytes, uint64_t* arena_stride);

static_assert(sizeof(bt2g_mm1_hit) == sizeof(Mm1Hit) && offsetof(bt2g_mm1_hit, score) == offsetof(Mm1Hit, score) && offsetof(bt2g_mm1_hit, epos) == offsetof(Mm1Hit, epos) &&
              offsetof(bt2g_mm1_hit, echr) == offsetof(Mm1Hit, echr) && offsetof(bt2g_mm1_hit, eqchr) == offsetof(Mm1Hit, eqchr), "bt2g_mm1_hit is the kernels' Mm1Hit");

int bt2g_one_mm_search(bt2g_ctx* c, const bt2g_reads* reads, const bt2g_read_params* d_rparams, const bt2g_align_params* params,
                       const bt2g_sweep_out* d_sweep, uint32_t cap, bt2g_mm1_hit* d_hits, uint8_t* d_n, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	if (!reads || !d_rparams || !params || !d_sweep || !d_hits || !d_n || cap == 0 || cap > 254) return fail(c, BT2G_ERR_ARG, "bad argument");
	hipStream_t st = (hipStream_t)stream;
	const uint64_t n = reads->n_reads;
	if (n == 0) return 0;
	// scratch of the four kernels: per-list hit counters, the queue of deferred branches, the task list, the two queue counters
	auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
	const uint64_t b_cnt = al(n * 4 * sizeof(unsigned int));
	const uint64_t qcap64 = n * 16 < 0xfffffff0ull ? n * 16 : 0xfffffff0ull;
	const uint64_t b_q = al(qcap64 * one_mm_task_bytes(c->off_size));
	const uint64_t b_t = al(n * 4 * sizeof(uint32_t));
	uint8_t* d_tmp = nullptr;
	hipError_t e = hipMalloc((void**)&d_tmp, b_cnt + b_q + b_t + 256);
	if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(one_mm scratch)");
	unsigned int* d_cntl = (unsigned int*)d_tmp;
	void* d_q = d_tmp + b_cnt;
	uint32_t* d_t = (uint32_t*)(d_tmp + b_cnt + b_q);
	unsigned int* d_qc = (unsigned int*)(d_tmp + b_cnt + b_q + b_t);
	e = (c->off_size == 4)
		? launch_one_mm(c->ix32, *params, *reads, d_rparams, d_sweep, cap, (void*)d_hits, d_n, d_cntl, d_q, (uint32_t)qcap64, d_qc, d_t, c->d_cnt, st)
		: launch_one_mm(c->ix64, *params, *reads, d_rparams, d_sweep, cap, (void*)d_hits, d_n, d_cntl, d_q, (uint32_t)qcap64, d_qc, d_t, c->d_cnt, st);
	const hipError_t e2 = hipStreamSynchronize(st);
	(void)hipFree(d_tmp);
	if (e != hipSuccess) return hip_fail(c, e, "k_one_mm");
	return e2 == hipSuccess ? 0 : hip_fail(c, e2, "k_one_mm");
}

int bt2g_index_rows(bt2g_ctx* c, uint64_t first_row, uint64_t n_rows, uint64_t* d_out, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	const uint64_t len = c->off_size == 4 ? (uint64_t)c->ix32.fw.len : (uint64_t)c->ix64.fw.len;
	if (!d_out || first_row > len || n_rows > len + 1 - first_row) return fail(c, BT2G_ERR_ARG, "bad argument");
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = (c->off_size == 4) ? launch_index_rows(c->ix32, first_row, n_rows, d_out, st) : launch_index_rows(c->ix64, first_row, n_rows, d_out, st);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_index_rows");
}

int bt2g_resolve_offsets(bt2g_ctx* c, const uint64_t* d_rows, const uint32_t* d_qlen, uint64_t n, int reject_straddle,
                         bt2g_resolved* d_out, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	if ((!d_rows || !d_qlen || !d_out) && n) return fail(c, BT2G_ERR_ARG, "bad argument");
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = (c->off_size == 4)
		? launch_resolve_offsets(c->ix32, d_rows, d_qlen, n, reject_straddle, d_out, c->d_cnt, st)
		: launch_resolve_offsets(c->ix64, d_rows, d_qlen, n, reject_straddle, d_out, c->d_cnt, st);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_resolve_offsets");
}

void bt2g_scoring_default(bt2g_scoring* sc) {
	if (!sc) return;
	sc->match_bonus = 0; sc->mm_pen_type = 3; sc->mm_max = 6; sc->mm_min = 2; sc->n_pen = 1;
	sc->rd_gap_const = 5; sc->rd_gap_linear = 3; sc->rf_gap_const = 5; sc->rf_gap_linear = 3; sc->gapbar = 4;
}

uint64_t bt2g_dp_out_bytes(uint32_t kind, uint32_t rows, uint32_t cols) {
	uint64_t b = sizeof(bt2g_dp_out);
	if (kind == BT2G_DP_EE_U8 || kind == BT2G_DP_EE_I16_BAND) b += (uint64_t)((cols + 3u) & ~3u) * 2 + pred_cells(rows ? rows : 1, cols ? cols : 1);      // the widest band the problem can have
	else if (kind == BT2G_DP_LOCAL) b += (uint64_t)rows * cols;      // one predecessor byte per cell
	else b += (uint64_t)rows * cols * 3 * 4;
	return (b + 7) & ~(uint64_t)7;
}

int bt2g_dp_fill(bt2g_ctx* c, const bt2g_scoring* sc, const bt2g_dp_problem* d_probs, uint32_t n,
                 const uint8_t* d_rd, const uint8_t* d_qu, const uint8_t* d_rf, uint8_t* d_out, void* stream) {
	if (!c) return BT2G_ERR_ARG;
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	if (!sc || (!d_probs && n) || (!d_out && n)) return fail(c, BT2G_ERR_ARG, "bad argument");
	if (n == 0) return 0;
	hipStream_t st = (hipStream_t)stream;
	// the per-wave scratch is sized from the problem shapes (this entry point is the inspection form of the fills; the worker sizes
	// its scratch once per batch)
	std::vector<bt2g_dp_problem> hp(n);
	hipError_t e = hipMemcpyAsync(hp.data(), d_probs, sizeof(bt2g_dp_problem) * n, hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) return hip_fail(c, e, "copy DP problems");
	uint32_t max_rows = 1, max_cols = (uint32_t)kMaxCols;
	for (const auto& p : hp) {
		if (p.rows == 0 || p.cols == 0 || p.rows > (uint32_t)kMaxLen || p.cols + 1 > (uint32_t)kMaxColsWide) return fail(c, BT2G_ERR_UNSUPPORTED, "DP problem outside 1..512 rows x 1..2175 columns");
		if (p.cols + 1 > max_cols) max_cols = (uint32_t)kMaxColsWide;       // (wide windows: the launch holds more per-column state in LDS)
		if (p.kind > BT2G_DP_EE_I16_BAND || (p.out_off & 7)) return fail(c, BT2G_ERR_ARG, "bad DP problem (kind / output offset)");
		if (p.rows > max_rows) max_rows = p.rows;
	}
	uint64_t mat_bytes, mask_bytes, pmask_bytes, stride;
	align_scratch_sizes(max_rows, true, 255, max_cols, mat_bytes, mask_bytes, pmask_bytes, stride);      // "paired": windows up to max_cols columns
	stride = (mat_bytes + mask_bytes + pmask_bytes + 4095) & ~(uint64_t)4095;
	uint32_t n_waves = c->n_cu * 4;
	if (n_waves > n) n_waves = n;
	const uint64_t need = stride * n_waves;
	if (need > c->dp_scratch_bytes) {
		if (c->d_dp_scratch) (void)hipFree(c->d_dp_scratch);
		c->d_dp_scratch = nullptr; c->dp_scratch_bytes = 0;
		e = hipMalloc((void**)&c->d_dp_scratch, need);
		if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(dp scratch)");
		c->dp_scratch_bytes = need;
	}
	e = hipMemsetAsync(c->d_dp_scratch, 0, need, st);      // epoch words and tagged masks start from zero
	if (e != hipSuccess) return hip_fail(c, e, "hipMemsetAsync(dp scratch)");
	AlignParams P;
	memset(&P, 0, sizeof(P));
	P.mm_type = sc->mm_pen_type; P.mm_max = sc->mm_max; P.mm_min = sc->mm_min; P.n_pen = sc->n_pen;
	P.rdgapo = sc->rd_gap_const + sc->rd_gap_linear; P.rdgape = sc->rd_gap_linear; P.rfgapo = sc->rf_gap_const + sc->rf_gap_linear; P.rfgape = sc->rf_gap_linear;
	P.gapbar = sc->gapbar; P.match_bonus = sc->match_bonus;
	e = launch_dp_fill(P, d_probs, n, d_rd, d_qu, d_rf, d_out, c->d_dp_scratch, stride, mat_bytes, mask_bytes, pmask_bytes, n_waves, max_cols, st);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_dp_fill");
}

uint64_t bt2g_align_result_stride(uint32_t khits) {
	if (khits == 0) khits = 1;
	// room for the slots of whichever worker class the batch runs in: the long-read class's are larger (khits <= 64 there), include/bt2g.h
	const uint64_t slot = khits > 64 ? sizeof(bt2g_aln) : (offsetof(bt2g_aln, ned) + (uint64_t)BT2G_MAX_EDITS_LONG * sizeof(bt2g_edit) + 7) & ~(uint64_t)7;
	const uint64_t b = offsetof(bt2g_read_result, alns) + (uint64_t)khits * slot;
	return (b + 15) & ~(uint64_t)15;
}

int bt2g_align_batch(bt2g_ctx* c, const bt2g_reads* reads, const bt2g_read_params* d_rparams,
                     const bt2g_align_params* params, uint32_t max_read_len, void* d_results, void* stream) {
	int rc = need_loaded(c);
	if (rc) return rc;
	if (!reads || !params || (!d_rparams && reads->n_reads) || (!d_results && reads->n_reads)) return fail(c, BT2G_ERR_ARG, "bad argument");
	if (params->khits < 1 || params->khits > BT2G_MAX_KHITS || (uint32_t)params->khits > bt2g_bk_max_alns()) return fail(c, BT2G_ERR_UNSUPPORTED, "-k outside [1,1000]");
	const bool bigk = params->khits > 64;      // the many-alignments class: its own per-wave capacities (and arena), see bt2g_align.hpp at BT2G_CLASS_BIG_K
	if (params->match_bonus < 0) return fail(c, BT2G_ERR_ARG, "negative match bonus");
	if (params->maxhalf < 0 || params->maxhalf > 255 || params->gapbar < 1 || params->rdgapo < 0 || params->rdgape < 0 || params->rfgapo < 0 || params->rfgape < 0 ||
	    params->max_dp_streak < 0 || params->n_seed_rounds < 0 || params->seed_mms < 0 || params->seed_mms > 1)
		return fail(c, BT2G_ERR_ARG, "alignment parameter out of range (maxhalf 0..255, gapbar >= 1, gap penalties >= 0, -N 0/1)");
	if (max_read_len == 0) max_read_len = 1;
	if (reads->n_reads == 0) return 0;
	if (max_read_len > BT2G_MAX_READ_LEN) max_read_len = BT2G_MAX_READ_LEN;      // (a longer read is flagged by the kernel, record by record)
	const bool longr = max_read_len > (uint32_t)kMaxLen;      // the long-read class
	if (longr && bigk) return fail(c, BT2G_ERR_UNSUPPORTED, "reads longer than 512 bp with -k above 64 (or -a) are not supported by this build");
	hipStream_t st = (hipStream_t)stream;
	// the working set of this stream (created on the stream's first batch)
	int si = -1;
	{
		std::lock_guard<std::mutex> g(c->slot_mu);
		for (int i = 0; i < bt2g_ctx::kMaxSlots && si < 0; i++) if (c->slots[i].used && c->slots[i].owner == st) si = i;
		for (int i = 0; i < bt2g_ctx::kMaxSlots && si < 0; i++) if (!c->slots[i].used) { si = i; c->slots[i].used = true; c->slots[i].owner = st; }
		if (si < 0) {
			// more streams than working sets (a caller that rotates through a stream pool, or recreates its streams): the set that has been
			// idle longest changes hands, once its last batch is done -- its buffers stay, its kernel timings are the old owner's and are dropped
			// (a set another thread is inside bt2g_align_batch with is never taken: ADVICE r5)
			for (int i = 0; i < bt2g_ctx::kMaxSlots; i++) if (!c->slots[i].busy && (si < 0 || c->slots[i].last_use < c->slots[si].last_use)) si = i;
			if (si < 0) return fail(c, BT2G_ERR_UNSUPPORTED, "every working set of the context is inside bt2g_align_batch: more than 4 streams at once");
			bt2g_ctx::BatchSlot& V = c->slots[si];
			if (V.fence_pending) { if (hipStreamSynchronize(V.owner) != hipSuccess) return fail(c, BT2G_ERR_HIP, "hipStreamSynchronize(working set changing streams)"); }
			else if (V.ev_valid && hipEventSynchronize(V.ev[6]) != hipSuccess) return fail(c, BT2G_ERR_HIP, "hipEventSynchronize(working set changing streams)");
			V.ev_valid = false; V.fence_pending = false; V.owner = st;
		}
		c->slots[si].last_use = ++c->n_batches;
		c->slots[si].busy++;
		c->last_slot = si;
	}
	bt2g_ctx::BatchSlot& S = c->slots[si];
	// leaves the slot on every return path; a return before the last kernel was queued (ev_valid still false) leaves work of unknown extent on
	// the stream, which is synchronised before the set may change owner
	struct SlotGuard { bt2g_ctx* c; bt2g_ctx::BatchSlot& S; ~SlotGuard() { std::lock_guard<std::mutex> g(c->slot_mu); S.busy--; if (!S.ev_valid) S.fence_pending = true; } } slot_guard{c, S};
	uint64_t mat_bytes, mask_bytes, pmask_bytes, arena_stride;
	if (params->paired && (reads->n_reads & 1u)) return fail(c, BT2G_ERR_ARG, "paired mode needs an even number of reads (mates interleaved)");
	uint32_t max_cols = dp_cols_for(*params);      // DP columns this launch holds (bt2g_align_params::max_dp_cols)
	// Occupancy class.  An end-to-end batch of unpaired reads needs DP windows of rows + 4 x maxhalf + 1 columns at most (seed extension only,
	// dp_framer.cpp:81-129): with that little per-column state in dynamic LDS, 20 waves per CU fit, and the worker compiled for 96 registers
	// (5 waves per SIMD) hides more of the latency its instruction stream is made of.  Local batches (2 KB of radix counters), pairs (wide
	// opposite-mate windows, a second matrix) and BT2G_NO_W5=1 stay with the 128-register build.
	// The 96-register build is also the SHORT-READ class: its capacities (longest read, seed positions per strand) are cut so that the LDS this
	// frees holds the backtrace's on-chip state; a batch goes there only when the class holds all of it, which needs the batch's widest seed table.
	hipError_t e;
	if (!S.d_next) {
		e = hipMalloc((void**)&S.d_next, 1024);
		if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(queue head)");
		(void)hipMemset(S.d_next, 0, 1024);
	}
	unsigned int max_seeds = params->max_seeds > 0 ? (unsigned int)params->max_seeds : 0u;
	if (max_seeds == 0 && c->precomp) {
		// no bound from the caller: count on the device and wait for the number (the only host-device round trip of a batch)
		e = launch_max_seeds(*reads, d_rparams, S.d_next + 8, st);
		if (e == hipSuccess) e = hipMemcpyAsync(&max_seeds, S.d_next + 8, sizeof(max_seeds), hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e != hipSuccess) return hip_fail(c, e, "k_max_seeds");
		if (max_seeds < 1) max_seeds = 1;
	}
	static const bool no_w5 = getenv("BT2G_NO_W5") != nullptr;
	static const uint32_t kStaticLdsBytes = bt2g_w5_static_lds();
	bool w5 = false;
	if (!no_w5 && !bigk && !longr && !params->paired && params->match_bonus == 0 && params->max_dp_cols <= kMaxCols && bt2g_w5_work_bytes() <= sizeof(Work) &&
	    max_read_len <= bt2g_w5_max_len() && max_seeds >= 1 && max_seeds <= bt2g_w5_max_offs()) {
		const uint32_t need = max_read_len + 4u * (uint32_t)(params->maxhalf > 0 ? params->maxhalf : 0) + 1u + 4u;
		const uint32_t lds_per_wave = (160u * 1024u) / (4u * 5u);
		if (need <= (uint32_t)kMaxCols && kStaticLdsBytes < lds_per_wave && kStaticLdsBytes + hot_tail_bytes(need, false) <= lds_per_wave) { w5 = true; max_cols = need; }      // (kStaticLdsBytes is 0xffffffff when the runtime would not say)
	}
	if (longr) {
		// a seed extension window of a long read is rows + 4 x maxhalf + 1 columns: more than the default launch holds
		const uint32_t need = max_read_len + 4u * (uint32_t)(params->maxhalf > 0 ? params->maxhalf : 0) + 1u + 4u;
		if (need > max_cols) max_cols = need < (uint32_t)kMaxColsWide ? need : (uint32_t)kMaxColsWide;
		bt2g_lr_scratch_sizes(max_read_len, params->paired != 0, (uint32_t)params->maxhalf, max_cols, &mat_bytes, &mask_bytes, &pmask_bytes, &arena_stride);
	} else if (bigk) bt2g_bk_scratch_sizes(max_read_len, params->paired != 0, (uint32_t)params->maxhalf, max_cols, &mat_bytes, &mask_bytes, &pmask_bytes, &arena_stride);
	else if (w5) bt2g_w5_scratch_sizes(max_read_len, 0, (uint32_t)params->maxhalf, max_cols, &mat_bytes, &mask_bytes, &pmask_bytes, &arena_stride);      // (its own work area and DP scratch: no pair state, band matrices only)
	else align_scratch_sizes(max_read_len, params->paired != 0, (uint32_t)params->maxhalf, max_cols, mat_bytes, mask_bytes, pmask_bytes, arena_stride);
	{ static const char* skew = getenv("BT2G_ARENA_SKEW");      // measurement knob: extra bytes per wave (do the waves' working sets meet in the same memory channels?)
	  if (skew) { arena_stride += (uint64_t)atoll(skew); if (getenv("BT2G_DEBUG_OCC")) fprintf(stderr, "[bt2g] arena stride %llu B (%llu x 4 KB)\n", (unsigned long long)arena_stride, (unsigned long long)(arena_stride >> 12)); } }
	// persistent waves (one read at a time each) pull reads from a device-side queue
	// (the short-read class is built for 20 waves per CU and runs 18: the worker is bound by what a CU can issue, not by latency -- round 6 measured
	// 293 / 285 / 286 / 296 ms per 2 M reads with 20 / 18 / 16 / 14 waves per CU resident, profiles/r06t_* -- and the LDS of the two waves left out goes to the others)
	static const uint32_t w5_waves = []() -> uint32_t { const char* v = getenv("BT2G_W5_WAVES"); const int k = v ? atoi(v) : 18; return (uint32_t)(k >= 4 && k <= (int)bt2g_w5_waves_per_cu() ? k : (int)bt2g_w5_waves_per_cu()); }();
	// LDS per wave of that class: its share of the CU's 160 KB, or less (BT2G_W5_LDS) -- what it leaves is where the FM kernels' blocks of the next batch fit
	static const uint32_t w5_lds = []() -> uint32_t { const char* v = getenv("BT2G_W5_LDS"); const uint32_t full = ((160u * 1024u) / w5_waves) & ~15u; const int k = v ? atoi(v) : 0; return (k >= 6144 && (uint32_t)k <= full) ? ((uint32_t)k & ~15u) : full; }();
	uint32_t n_waves = c->n_cu * (w5 ? w5_waves : longr ? bt2g_lr_waves_per_cu() : bigk ? bt2g_bk_waves_per_cu() : align_waves_per_cu());
	{ static const int pct = getenv("BT2G_WAVES_PCT") ? atoi(getenv("BT2G_WAVES_PCT")) : 100;      // measurement knob: launch a fraction of the resident waves (latency- or throughput-bound?)
	  if (pct > 0 && pct < 100) n_waves = (uint32_t)((uint64_t)n_waves * (uint32_t)pct / 100u); if (n_waves == 0) n_waves = 1; }
	if (n_waves > reads->n_reads) n_waves = reads->n_reads;
	uint64_t need = arena_stride * n_waves;
	if (need > S.arena_bytes && S.arena_cut_stride == arena_stride && S.arena_bytes / arena_stride >= 64) {
		// the arena was cut to the memory budget for launches of this shape: run with the waves it holds instead of allocating again
		n_waves = (uint32_t)(S.arena_bytes / arena_stride);
		need = arena_stride * n_waves;
	}
	if (need > S.arena_bytes) {
		// Every stream's working set has its own arena (Work + DP matrices per resident wave): with wide opposite-mate windows and long mates a
		// wave's share reaches tens of MB, and three sets of 4 096 of them would not fit next to the index.  The launch is cut to the waves
		// whose arena fits in 90 % of what is free (fewer resident waves: slower, never wrong) before the allocation is given up.
		if (S.d_arena) (void)hipFree(S.d_arena);
		S.d_arena = nullptr; S.arena_bytes = 0; S.arena_layout = 0; S.arena_cut_stride = 0;
		size_t free_b = 0, total_b = 0;
		// (... and to a fifth of the device: the working sets of the other streams need theirs)
		uint64_t budget = 0;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>((uint64_t)(free_b * 0.9), (uint64_t)(total_b * 0.22));
		if (budget && need > budget) {
			const uint64_t fit = budget / arena_stride;
			if (fit < 64) return fail(c, BT2G_ERR_HIP, "not enough device memory for the worker arena of this batch (reads / mate windows this long need more per wave than is free)");
			n_waves = (uint32_t)fit;
			need = arena_stride * n_waves;
			S.arena_cut_stride = arena_stride;
		}
		e = hipMalloc((void**)&S.d_arena, need);
		if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(worker arena)");
		S.arena_bytes = need;
	}
	// the epoch-tagged backtrace masks live in the arena across launches: (re)start from zero whenever its layout changes
	const uint64_t layout = arena_stride ^ (mat_bytes << 1) ^ (mask_bytes << 2) ^ (pmask_bytes << 3) ^ (params->paired ? 1ull << 63 : 0) ^ (w5 ? 1ull << 62 : 0) ^ (bigk ? 1ull << 61 : 0) ^ (longr ? 1ull << 60 : 0);
	if (layout != S.arena_layout) {
		e = hipMemsetAsync(S.d_arena, 0, S.arena_bytes, st);
		if (e != hipSuccess) return hip_fail(c, e, "hipMemsetAsync(worker arena)");
		S.arena_layout = layout;
	}
	// The pure FM phases of every read (exact sweep, 1-mismatch search, seed round 0 and its seed-hit
	// extension) run first as lane-per-task kernels; the per-read worker then consumes their output.
	PreComp pre;
	memset(&pre, 0, sizeof(pre));
	if (!S.ev[0]) for (int i = 0; i < 7; i++) if (hipEventCreate(&S.ev[i]) != hipSuccess) return fail(c, BT2G_ERR_HIP, "hipEventCreate");
	{ std::lock_guard<std::mutex> g(c->slot_mu); S.ev_valid = false; }
	auto mark = [&](int i) { (void)hipEventRecord(S.ev[i], st); };
	mark(0);
	if (c->precomp) {
		if (max_seeds < 1) max_seeds = 1;
		{ const uint32_t offs_cap = longr ? bt2g_lr_max_offs() : 64u; if (max_seeds > offs_cap) max_seeds = offs_cap; }       // kMaxOffs of the class: longer seed lists are flagged by the worker
		const uint32_t cap = 8;
		const uint64_t n = reads->n_reads;
		auto al = [](uint64_t v) { return (v + 255) & ~255ull; };
		const uint64_t b_sweep = al(n * sizeof(bt2g_sweep_out));
		const uint64_t b_seeds = al(n * 2 * max_seeds * sizeof(bt2g_seed_hit));
		const uint64_t b_ext = al(n * 2 * max_seeds * sizeof(uint32_t));
		const uint64_t b_joff = al(n * 2 * max_seeds * sizeof(uint64_t));
		const uint64_t b_mm1 = al(n * 4 * cap * sizeof(Mm1Hit));
		const uint64_t b_mm1n = al(n * 4);
		// 1-mismatch search: per-list hit counters and the queue of deferred branches (16 per read on average; beyond that
		// the scan kernel follows branches itself)
		const uint64_t b_mm1c = al(n * 4 * sizeof(unsigned int));
		const uint64_t qcap64 = n * 16 < 0xfffffff0ull ? n * 16 : 0xfffffff0ull;
		const uint64_t b_mm1q = al(qcap64 * one_mm_task_bytes(c->off_size));
		const uint64_t b_mm1t = al(n * 4 * sizeof(uint32_t));      // the scan's task list: the (read, strand, direction) combinations that are searched at all
		// re-seeding rounds are pre-computed too (round 6: for pairs as well -- a pair whose mates both pass the filters has half as many rounds, seed_rounds_of)
		uint32_t pre_rounds = 1;
		if (params->seed_mms == 0 && params->n_seed_rounds > 1) pre_rounds = (uint32_t)params->n_seed_rounds < kMaxPreRounds ? (uint32_t)params->n_seed_rounds : kMaxPreRounds;
		const uint64_t tot = b_sweep + (b_seeds + b_ext + b_joff) * pre_rounds + b_mm1 + b_mm1n + b_mm1c + b_mm1q + b_mm1t;
		if (tot > S.pre_bytes) {
			if (S.d_pre) (void)hipFree(S.d_pre);
			S.d_pre = nullptr; S.pre_bytes = 0;
			e = hipMalloc((void**)&S.d_pre, tot);
			if (e != hipSuccess) return hip_fail(c, e, "hipMalloc(pre-computation)");
			S.pre_bytes = tot;
		}
		uint8_t* p = S.d_pre;
		bt2g_sweep_out* d_sweep = (bt2g_sweep_out*)p; p += b_sweep;
		bt2g_seed_hit* d_seeds = (bt2g_seed_hit*)p; p += b_seeds;
		uint64_t* d_joff = (uint64_t*)p; p += b_joff;
		uint32_t* d_ext = (uint32_t*)p; p += b_ext;
		Mm1Hit* d_mm1 = (Mm1Hit*)p; p += b_mm1;
		uint8_t* d_mm1n = p;
		const bool s = c->off_size == 4;
		mark(0);
		if (params->do_exact_upfront) {
			e = s ? launch_exact_sweep(c->ix32, *reads, params->nofw, params->norc, 2, d_sweep, c->d_cnt, st)
			      : launch_exact_sweep(c->ix64, *reads, params->nofw, params->norc, 2, d_sweep, c->d_cnt, st);
			if (e != hipSuccess) return hip_fail(c, e, "k_exact_sweep");
			pre.sweep = (decltype(pre.sweep))d_sweep;
			mark(1);
			if (params->do_1mm_upfront) {
				unsigned int* d_mm1c = (unsigned int*)(d_mm1n + b_mm1n);
				void* d_mm1q = d_mm1n + b_mm1n + b_mm1c;
				uint32_t* d_mm1t = (uint32_t*)(d_mm1n + b_mm1n + b_mm1c + b_mm1q);
				e = s ? launch_one_mm(c->ix32, *params, *reads, d_rparams, d_sweep, cap, d_mm1, d_mm1n, d_mm1c, d_mm1q, (uint32_t)qcap64, S.d_next + 12, d_mm1t, c->d_cnt, st)
				      : launch_one_mm(c->ix64, *params, *reads, d_rparams, d_sweep, cap, d_mm1, d_mm1n, d_mm1c, d_mm1q, (uint32_t)qcap64, S.d_next + 12, d_mm1t, c->d_cnt, st);
				if (e != hipSuccess) return hip_fail(c, e, "k_one_mm");
				pre.mm1 = (decltype(pre.mm1))d_mm1; pre.mm1_n = (decltype(pre.mm1_n))d_mm1n;
			}
		} else mark(1);
		mark(2);
		if (params->seed_mms == 0) {   // -N 1 seeds are searched by the worker itself (several ranges per seed)
			e = s ? launch_seed_search_exact(c->ix32, *reads, nullptr, nullptr, nullptr, d_rparams, max_seeds, d_seeds, c->d_cnt, st)
			      : launch_seed_search_exact(c->ix64, *reads, nullptr, nullptr, nullptr, d_rparams, max_seeds, d_seeds, c->d_cnt, st);
			if (e != hipSuccess) return hip_fail(c, e, "k_seed_search_exact");
			pre.seeds = (decltype(pre.seeds))d_seeds;
			mark(3);
			if (params->do_extend) {
				e = s ? launch_extend_hits(c->ix32, *reads, d_rparams, max_seeds, (params->do_extend & 2) ? 0 : 1, d_seeds, d_ext, d_joff, c->d_cnt, st)
				      : launch_extend_hits(c->ix64, *reads, d_rparams, max_seeds, (params->do_extend & 2) ? 0 : 1, d_seeds, d_ext, d_joff, c->d_cnt, st);
				if (e != hipSuccess) return hip_fail(c, e, "k_extend_hits");
				pre.ext = (decltype(pre.ext))d_ext; pre.joff = (decltype(pre.joff))d_joff;
			}
			// rounds 1..: the same kernels on the shifted seeds, for the reads repetitive enough to be re-seeded
			const bt2g_seed_hit* prev = d_seeds;
			uint8_t* q = d_mm1n + b_mm1n + b_mm1c + b_mm1q + b_mm1t;
			for (uint32_t ri = 1; ri < pre_rounds; ri++) {
				bt2g_seed_hit* sr = (bt2g_seed_hit*)q; q += b_seeds;
				uint64_t* jr = (uint64_t*)q; q += b_joff;
				uint32_t* er = (uint32_t*)q; q += b_ext;
				ReseedCtl ctl; ctl.prev = prev; ctl.n_seed_rounds = (uint32_t)params->n_seed_rounds; ctl.boost_thresh = (uint32_t)params->seed_boost_thresh; ctl.nofw = params->nofw; ctl.norc = params->norc; ctl.paired = params->paired;
				e = s ? launch_seed_search_exact(c->ix32, *reads, nullptr, nullptr, nullptr, d_rparams, max_seeds, sr, c->d_cnt, st, ri, &ctl)
				      : launch_seed_search_exact(c->ix64, *reads, nullptr, nullptr, nullptr, d_rparams, max_seeds, sr, c->d_cnt, st, ri, &ctl);
				if (e != hipSuccess) return hip_fail(c, e, "k_seed_search_exact (re-seed)");
				pre.seeds_r[ri] = (decltype(pre.seeds_r[ri]))sr;
				if (params->do_extend) {
					e = s ? launch_extend_hits(c->ix32, *reads, d_rparams, max_seeds, (params->do_extend & 2) ? 0 : 1, sr, er, jr, c->d_cnt, st, ri, (uint32_t)params->n_seed_rounds, params->paired)
					      : launch_extend_hits(c->ix64, *reads, d_rparams, max_seeds, (params->do_extend & 2) ? 0 : 1, sr, er, jr, c->d_cnt, st, ri, (uint32_t)params->n_seed_rounds, params->paired);
					if (e != hipSuccess) return hip_fail(c, e, "k_extend_hits (re-seed)");
					pre.ext_r[ri] = (decltype(pre.ext_r[ri]))er; pre.joff_r[ri] = (decltype(pre.joff_r[ri]))jr;
				}
				prev = sr;
			}
		} else { mark(3); }
		pre.max_seeds = max_seeds; pre.mm1_cap = cap;
		mark(4);
	} else { for (int i = 1; i <= 4; i++) mark(i); }
	mark(5);
	const uint64_t stride = bt2g_align_result_stride((uint32_t)params->khits);
	if (w5)
		e = bt2g_w5_launch_align(c->off_size, c->off_size == 4 ? (const void*)&c->ix32 : (const void*)&c->ix64, params, reads, d_rparams, (uint8_t*)d_results, stride, S.d_arena, arena_stride,
		                         mat_bytes, mask_bytes, pmask_bytes, n_waves, S.d_next, (unsigned long long*)(S.d_next + 16), &pre, max_read_len, max_cols, w5_lds, st);
	else if (longr)
		e = bt2g_lr_launch_align(c->off_size, c->off_size == 4 ? (const void*)&c->ix32 : (const void*)&c->ix64, params, reads, d_rparams, (uint8_t*)d_results, stride, S.d_arena, arena_stride,
		                         mat_bytes, mask_bytes, pmask_bytes, n_waves, S.d_next, (unsigned long long*)(S.d_next + 16), &pre, max_read_len, max_cols, (160u * 1024u) / bt2g_lr_waves_per_cu(), st);
	else if (bigk)
		e = bt2g_bk_launch_align(c->off_size, c->off_size == 4 ? (const void*)&c->ix32 : (const void*)&c->ix64, params, reads, d_rparams, (uint8_t*)d_results, stride, S.d_arena, arena_stride,
		                         mat_bytes, mask_bytes, pmask_bytes, n_waves, S.d_next, (unsigned long long*)(S.d_next + 16), &pre, max_read_len, max_cols, (160u * 1024u) / bt2g_bk_waves_per_cu(), st);
	else
	e = (c->off_size == 4)
		? launch_align(c->ix32, *params, *reads, d_rparams, (uint8_t*)d_results, stride, S.d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, n_waves, S.d_next, (unsigned long long*)(S.d_next + 16), pre, max_read_len, max_cols, (160u * 1024u) / align_waves_per_cu(), st)
		: launch_align(c->ix64, *params, *reads, d_rparams, (uint8_t*)d_results, stride, S.d_arena, arena_stride, mat_bytes, mask_bytes, pmask_bytes, n_waves, S.d_next, (unsigned long long*)(S.d_next + 16), pre, max_read_len, max_cols, (160u * 1024u) / align_waves_per_cu(), st);
	if (e != hipSuccess) return hip_fail(c, e, "k_align_reads");
	mark(6);
	{ std::lock_guard<std::mutex> g(c->slot_mu); S.ev_valid = true; S.fence_pending = false; }
	return 0;
}

int bt2g_results_pack(bt2g_ctx* c, const void* d_results, uint32_t n_reads, uint32_t khits, void* d_packed, uint64_t* d_offsets, void* stream) {
	if (!c) return BT2G_ERR_ARG;
	if (!d_results || !d_packed || !d_offsets) return fail(c, BT2G_ERR_ARG, "bt2g_results_pack: null buffer");
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	if (khits == 0) khits = 1;
	hipError_t e = launch_pack_results(d_results, bt2g_align_result_stride(khits), n_reads, khits, d_packed, d_offsets, (hipStream_t)stream);
	return e == hipSuccess ? 0 : hip_fail(c, e, "k_pack_results");
}

static int timing_of_slot(bt2g_ctx* c, int si, float* out_ms5) {
	for (int i = 0; i < 5; i++) out_ms5[i] = 0.f;
	bool valid = false;
	if (si >= 0) { std::lock_guard<std::mutex> g(c->slot_mu); valid = c->slots[si].ev_valid; }
	if (!valid) return fail(c, BT2G_ERR_ARG, "no align batch has been launched");
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	bt2g_ctx::BatchSlot& S = c->slots[si];
	hipError_t e = hipEventSynchronize(S.ev[6]);
	if (e != hipSuccess) return hip_fail(c, e, "hipEventSynchronize");
	// k_exact_sweep, k_one_mm, k_seed_search_exact, k_extend_hits, k_align_reads
	const int a[5] = {0, 1, 2, 3, 5}, b[5] = {1, 2, 3, 4, 6};
	for (int i = 0; i < 5; i++) {
		e = hipEventElapsedTime(&out_ms5[i], S.ev[a[i]], S.ev[b[i]]);
		if (e != hipSuccess) return hip_fail(c, e, "hipEventElapsedTime");
	}
	return 0;
}
int bt2g_align_timing_read(bt2g_ctx* c, float* out_ms5) {
	if (!c || !out_ms5) return BT2G_ERR_ARG;
	int si;
	{ std::lock_guard<std::mutex> g(c->slot_mu); si = c->last_slot; }
	return timing_of_slot(c, si, out_ms5);
}
int bt2g_align_timing_read_on(bt2g_ctx* c, void* stream, float* out_ms5) {
	if (!c || !out_ms5) return BT2G_ERR_ARG;
	int si = -1;
	{
		std::lock_guard<std::mutex> g(c->slot_mu);
		for (int i = 0; i < bt2g_ctx::kMaxSlots; i++) if (c->slots[i].used && c->slots[i].owner == (hipStream_t)stream) si = i;
	}
	return timing_of_slot(c, si, out_ms5);
}

int bt2g_align_profile_read(bt2g_ctx* c, uint64_t* out32, int reset, void* stream) {
	if (!c || !out32) return BT2G_ERR_ARG;
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	hipStream_t st = (hipStream_t)stream;
	memset(out32, 0, 32 * 8);
	// every working set (stream) of the context has its own profile block: the sum is reported.  Call it with the batches done.
	for (auto& S : c->slots) {
		if (!S.d_next) continue;
		uint64_t tmp[32];
		hipError_t e = hipMemcpyAsync(tmp, S.d_next + 16, 32 * 8, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess && reset) e = hipMemsetAsync(S.d_next + 16, 0, 32 * 8, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e != hipSuccess) return hip_fail(c, e, "read profile");
		for (int i = 0; i < 32; i++) out32[i] += tmp[i];
	}
	return 0;
}

int bt2g_counters_read(bt2g_ctx* c, bt2g_counters* out, int reset, void* stream) {
	if (!c || !out) return BT2G_ERR_ARG;
	if (hipSetDevice(c->device) != hipSuccess) return BT2G_ERR_NO_DEVICE;
	hipStream_t st = (hipStream_t)stream;
	std::vector<DevCounters> hs(kCntSlots);
	hipError_t e = hipMemcpyAsync(hs.data(), c->d_cnt, kCntSlots * sizeof(DevCounters), hipMemcpyDeviceToHost, st);
	if (e == hipSuccess && reset) e = hipMemsetAsync(c->d_cnt, 0, kCntSlots * sizeof(DevCounters), st);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) return hip_fail(c, e, "read counters");
	DevCounters h; h.rank_queries = h.sa_lookups = h.ftab_lookups = h.dp_cells = h.bwops = 0;
	for (const DevCounters& k : hs) { h.rank_queries += k.rank_queries; h.sa_lookups += k.sa_lookups; h.ftab_lookups += k.ftab_lookups; h.dp_cells += k.dp_cells; h.bwops += k.bwops; }
	out->rank_queries = h.rank_queries; out->sa_lookups = h.sa_lookups; out->ftab_lookups = h.ftab_lookups;
	out->dp_cells = h.dp_cells; out->bwops = h.bwops;
	return 0;
}

} // extern "C"
