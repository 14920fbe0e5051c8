// bt2g_search.cpp -- the aligner driver behind `bowtie2-align-s` / `bowtie2-align-l`, as a library entry point:
//   extern "C" int bowtie(int argc, const char **argv)        -- the reference's bt2_search.cpp:5223
// (bt2g_main.cpp is the 10-line main() around it, as the reference's bowtie_main.cpp is around its own).
//
// Accepts the argv the reference's Perl wrapper execs (bowtie2:482: "--wrapper basic-0 ..."),
// reads FASTQ, runs the per-read worker on the GPU through the C ABI (include/bt2g.h) and
// writes SAM + the stderr summary in the reference's format.  Options this build does not implement
// are rejected with an error rather than silently approximated.  There is no CPU alignment path here.
// A fatal condition found before the pipeline starts (bad arguments, missing index, no gfx950 device, unreadable files)
// prints "Error: ..." and makes bowtie() return the exit status the executable would have had (see die()).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bt2g.h"
#include "bt2g_host.hpp"
#include "bt2g_pipeline.hpp"
#include "bt2g_cli.hpp"

using namespace bt2g;

struct CliExit { int code; };
static std::thread::id g_caller;      // the thread inside bowtie()
// Fatal condition.  On the calling thread (arguments, index, files, device set-up) it unwinds to bowtie(), which returns the
// status.  On one of the pipeline's own threads (malformed input met while parsing, a device error mid-run) there is nobody
// to unwind to while the other stages keep running: it ends the process, as the reference's worker threads do.
[[noreturn]] static void die(const std::string& msg, int code = 1) {
	fprintf(stderr, "Error: %s\n", msg.c_str());
	if (std::this_thread::get_id() != g_caller) { fflush(stdout); fflush(stderr); exit(code); }
	throw CliExit{code};
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) die(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct DevBuf {
	void* p = nullptr; size_t cap = 0;
	void ensure(size_t n) { if (n > cap) { if (p) (void)hipFree(p); HIP_OK(hipMalloc(&p, n)); cap = n; } }
	~DevBuf() { if (p) (void)hipFree(p); }
};

// Page-locked host buffers for the result records, recycled from batch to batch (a fresh pageable buffer per
// batch costs more in page faults and staging copies than the alignment itself).
class PinnedPool {
public:
	std::shared_ptr<void> get(size_t n) {
		void* p = nullptr; size_t cap = 0;
		{
			std::lock_guard<std::mutex> g(mu_);
			for (size_t i = 0; i < free_.size(); i++) if (free_[i].second >= n) { p = free_[i].first; cap = free_[i].second; free_.erase(free_.begin() + (long)i); break; }
			if (!p && !free_.empty()) { (void)hipHostFree(free_.back().first); free_.pop_back(); }   // too small: replace it
		}
		if (!p) { cap = n + n / 4 + (1u << 20); HIP_OK(hipHostMalloc(&p, cap, hipHostMallocPortable)); }
		return std::shared_ptr<void>(p, [this, cap](void* q) { std::lock_guard<std::mutex> g(mu_); free_.emplace_back(q, cap); });
	}
	~PinnedPool() { for (auto& f : free_) (void)hipHostFree(f.first); }
private:
	std::mutex mu_;
	std::vector<std::pair<void*, size_t>> free_;
};

static int run_search(int argc, char** argv) {
	Options opt;
	std::string cmdline;
	for (int i = 0; i < argc; i++) { if (i) cmdline.push_back(' '); cmdline += argv[i]; }
	opt.cmdline = cmdline;
	CliExtra ex;
	ex.allow_paired = true;
	{
		const std::string err = parse_cli(argc, argv, opt, ex);
		if (ex.arg_desc) { print_arg_desc(); return 0; }
		if (ex.version) { print_version(argv[0]); return 0; }
		if (ex.help) { print_usage(argv[0]); return 0; }
		if (!err.empty()) die(err, 1);
	}
	if (!ex.pg_cmdline.empty()) opt.cmdline = ex.pg_cmdline;
	const bool metrics = ex.metrics;
	unsigned long long n_flagged = 0;
	if (opt.index_base.empty() || (opt.reads_file.empty() && !opt.paired)) die("usage: bowtie2-align-s [options] -x <index> {-U <reads.fq> | -1 <m1.fq> -2 <m2.fq>} [-S out.sam]");

	// The read sources are opened first: a reads file that cannot be opened is reported before 28 GB of index go to the device.
	// -p: host threads for FASTQ parsing and SAM formatting (the alignment itself is on the device)
	const unsigned host_threads = opt.threads > 0 ? (unsigned)opt.threads : 1u;
	const bool inter = !opt.interleaved_file.empty();
	const std::string src1 = inter ? opt.interleaved_file : (opt.paired ? opt.mate1_file : opt.reads_file);
	FastqBatcher fq(src1, opt, host_threads);
	if (!fq.ok()) die(fq.open_error("cannot open reads file " + src1));
	std::unique_ptr<FastqBatcher> fq2;
	if (opt.paired && !inter) { fq2.reset(new FastqBatcher(opt.mate2_file, opt, host_threads)); if (!fq2->ok()) die(fq2->open_error("cannot open reads file " + opt.mate2_file)); fq.set_bam_mate(1); fq2->set_bam_mate(2); }
	if (ex.shard_bytes) {
		std::string e;
		if (!fq.set_range(ex.range_a[0], ex.range_b[0], ex.first_read, e)) die(e);
		if (fq2 && !fq2->set_range(ex.range_a[1], ex.range_b[1], ex.first_read, e)) die(e);
	}
	// -U next to -1/-2: the unpaired reads follow the pairs (same reader thread, batches marked paired or not one by one)
	std::unique_ptr<FastqBatcher> fq_unp;
	if (opt.mixed_unpaired) { fq_unp.reset(new FastqBatcher(opt.reads_file, opt, host_threads)); if (!fq_unp->ok()) die("cannot open reads file " + opt.reads_file); }
	// --gpu a[,b,...]: one context (full index replica) per listed device; read batches are dealt to whichever device
	// is free and the writer puts them back in input order (reads are independent: SURVEY.md 8e)
	std::vector<int> devices = ex.devices;
	if (devices.empty()) devices.push_back(0);
	const size_t ndev = devices.size();
	std::vector<bt2g_ctx*> ctxs(ndev, nullptr);
	// a fatal error on this thread unwinds to bowtie() (die()): the contexts (index replicas in HBM) and the output file go with it
	struct Cleanup {
		std::vector<bt2g_ctx*>& ctxs; FILE* out = nullptr; FILE* idx = nullptr;
		~Cleanup() { for (bt2g_ctx*& c : ctxs) if (c) { bt2g_ctx_destroy(c); c = nullptr; } if (out && out != stdout) fclose(out); if (idx) fclose(idx); }
	} cleanup{ctxs};
	auto t0 = std::chrono::steady_clock::now();
	{
		std::vector<std::thread> loaders;
		std::vector<std::string> errs(ndev);
		for (size_t d = 0; d < ndev; d++) loaders.emplace_back([&, d]() {
			if (bt2g_ctx_create(devices[d], &ctxs[d])) { errs[d] = "no usable MI355X (gfx950) device " + std::to_string(devices[d]) + " -- this build has no CPU alignment path"; return; }
			if (bt2g_index_load(ctxs[d], opt.index_base.c_str())) errs[d] = std::string("could not load index: ") + bt2g_last_error(ctxs[d]);
		});
		for (auto& t : loaders) t.join();
		for (const std::string& e : errs) if (!e.empty()) die(e, 1);
	}
	bt2g_index_info info;
	bt2g_index_info_get(ctxs[0], &info);
	auto t1 = std::chrono::steady_clock::now();
	RefInfo ref;
	for (uint64_t i = 0; i < info.n_pat; i++) {
		const char* nm; uint64_t ln;
		bt2g_index_refname(ctxs[0], i, &nm, &ln);
		ref.names.push_back(nm); ref.lens.push_back(ln);
	}
	FILE* out = opt.out_file.empty() ? stdout : fopen(opt.out_file.c_str(), "wb");
	if (!out) die("cannot open output " + opt.out_file);
	cleanup.out = out;
	// every write of the SAM stream is checked: a full disk ends the run with an error, not with a truncated file and exit status 0
	auto put = [&](const char* p_, size_t n_) { if (n_ && fwrite(p_, 1, n_, out) != n_) die(std::string("could not write SAM output: ") + strerror(errno)); };
	std::string o;
	if (!opt.sam_no_hd) sam_header(o, ref, opt.cmdline, true, !opt.sam_no_sq, opt.rg_id, opt.rgs);   // --no-hd drops every header line (bt2_search.cpp:5126-5130)
	put(o.data(), o.size());

	AlignParams P;
	opt.to_params(P, info.off_size == 8);
	const uint64_t stride = bt2g_align_result_stride((uint32_t)P.khits);
	// keep the result records of one batch within ~2 GB (a record holds up to -k alignments of 1.2 KB each)
	const size_t batch_reads = std::max<size_t>(2, std::min<size_t>(ex.batch_reads, (size_t)((2ull << 30) / stride)) & ~(size_t)1);   // even: pairs stay together

	PairSummary psumm;
	AlnSummary summ;
	std::mutex align_mu;
	double align_s = 0, t_format = 0, t_write = 0, t_h2d = 0, t_d2h = 0;
	double ts_first_parsed = -1, ts_first_aligned = -1, ts_last_parsed = 0, ts_last_aligned = 0;      // -t: seconds after the index load
	auto since_load = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count(); };
	typedef std::unique_ptr<HostBatch> BatchPtr;
	// finished batches go back to the reader with their memory (HostBatch::recycle)
	struct BatchPool {
		std::mutex mu; std::vector<BatchPtr> free_;
		BatchPtr get() { { std::lock_guard<std::mutex> g(mu); if (!free_.empty()) { BatchPtr b = std::move(free_.back()); free_.pop_back(); return b; } } return BatchPtr(new HostBatch()); }
		void put(BatchPtr b) {
			if (!b) return;
			BatchPtr m0 = std::move(b->mate_src[0]), m1 = std::move(b->mate_src[1]);
			b->recycle();
			std::lock_guard<std::mutex> g(mu);
			if (free_.size() < 24) free_.push_back(std::move(b));
			for (BatchPtr* m : { &m0, &m1 }) if (*m) { (*m)->recycle(); if (free_.size() < 24) free_.push_back(std::move(*m)); }
		}
	} pool;
	const size_t kWorkersPerDev = 3;
	BoundedQueue<BatchPtr> q_in(ndev * kWorkersPerDev + 1), q_out(ndev * kWorkersPerDev + 1);

	FILE* shard_idx = nullptr;
	if (!ex.shard_index.empty()) { shard_idx = fopen(ex.shard_index.c_str(), "w"); if (!shard_idx) die("cannot open " + ex.shard_index); cleanup.idx = shard_idx; }
	// Batch sizes.  Fixed (--batch, default 262 144 reads: with the three launches the device threads keep in flight the device runs within
	// 3 % of its rate on 2 M-read launches, profiles/r04af_*), or -- with --batch-max N, an experiment that did not pay on file-sized inputs --
	// varied: ramping up from 64 K reads, doubling to N, and, when the input is one plain file so that what is left of it can be estimated,
	// tapering off again towards the end (a run pays one batch of latency at either end).  The SAM does not depend on how the input is cut.
	const bool vary_batches = ex.batch_max > 0 && ex.shard_world <= 1 && !inter;
	const size_t batch_cap = std::max<size_t>(2, std::min<size_t>(std::max<size_t>(ex.batch_max, 1u << 16), (size_t)((2ull << 30) / stride)) & ~(size_t)1);
	const uint64_t input_bytes = vary_batches ? fq.plain_size() : 0;
	auto batch_size_for = [&](uint64_t k, uint64_t reads_done) -> size_t {
		if (!vary_batches) return batch_reads;
		size_t s = (size_t)65536 << (k < 8 ? k : 8);
		if (s > batch_cap) s = batch_cap;
		const uint64_t bytes_done = fq.bytes_read();
		if (input_bytes && reads_done > 0 && bytes_done > 0) {
			const uint64_t left = input_bytes > bytes_done ? input_bytes - bytes_done : 0;
			const uint64_t est = (uint64_t)((double)left * (double)(opt.paired ? 2 * reads_done : reads_done) / (double)bytes_done);      // (paired: this is mate 1's file, a batch holds both mates)
			size_t t = (size_t)((est / 3 + 16383) & ~(uint64_t)16383);
			if (t < 32768) t = 32768;
			if (t < s) s = t;
		}
		return s & ~(size_t)1;
	};
	std::thread reader([&]() {
		uint64_t seq = 0, blk = 0, reads_done = 0;
		bool unp_phase = false;          // mixed input: the pair sources are exhausted, the -U files are being read
		size_t cur_size = batch_size_for(0, 0);
		for (;;) {
			BatchPtr b = pool.get();
			// (the scan of the batch after this one starts inside next(): its size is decided now)
			const size_t next_size = batch_size_for(blk + 1, reads_done + cur_size / (opt.paired ? 2 : 1));
			if (unp_phase) fq_unp->next(*b, batch_reads, (size_t)BT2G_MAX_READ_LEN);
			else
			if (inter) { fq.next(*b, batch_reads & ~(size_t)1, (size_t)BT2G_MAX_READ_LEN); finalize_interleaved(*b, opt); }
			else if (opt.paired) {
				// one batch per mate file in lockstep, interleaved into a batch of pairs
				BatchPtr b1 = pool.get(), b2 = pool.get();
				fq.next(*b1, cur_size / 2, (size_t)BT2G_MAX_READ_LEN, next_size / 2);
				fq2->next(*b2, cur_size / 2, (size_t)BT2G_MAX_READ_LEN, next_size / 2);
				merge_mate_batches(std::move(b1), std::move(b2), *b, opt);
			} else
			fq.next(*b, cur_size, (size_t)BT2G_MAX_READ_LEN, next_size);
			reads_done += b->reads.size() / (opt.paired ? 2 : 1);
			cur_size = next_size;
			if (fq_unp && !unp_phase && b->last && !b->upto_hit && b->bad_input.empty() && b->too_long.empty()) { b->last = false; unp_phase = true; }   // the run goes on with the unpaired reads
			// --shard r/N: batch k is block k of the input; this rank keeps blocks r, r+N, ... (an emptied batch still carries
			// the end-of-input marker and any input error)
			b->block_id = blk++;
			if (ex.shard_world > 1 && (int)(b->block_id % (uint64_t)ex.shard_world) != ex.shard_rank) {
				if (!b->last && b->bad_input.empty()) continue;
				b->clear_reads();
			}
			b->seqno = seq++;
			const bool last = b->last;
			{ std::lock_guard<std::mutex> g2(align_mu); const double ts = since_load(); if (ts_first_parsed < 0) ts_first_parsed = ts; ts_last_parsed = ts; }
			q_in.push(std::move(b));
			if (last) break;
		}
		for (size_t d = 0; d < ndev * kWorkersPerDev; d++) { BatchPtr stop(new HostBatch()); stop->terminator = true; q_in.push(std::move(stop)); }
	});
	std::thread writer([&]() {
		std::vector<std::string> parts;
		std::map<uint64_t, BatchPtr> pending;       // batches that finished ahead of their turn
		uint64_t next_seq = 0;
		bool done = false;
		while (!done) {
			BatchPtr got = q_out.pop();
			pending[got->seqno] = std::move(got);
			while (!pending.empty() && pending.begin()->first == next_seq) {
				BatchPtr b = std::move(pending.begin()->second);
				pending.erase(pending.begin());
				next_seq++;
				const auto tf0_ = std::chrono::steady_clock::now();
				BatchTally tally;
				format_batch(*b, opt, ref, host_threads, parts, tally);
				summ.merge(tally.summ); psumm.merge(tally.psumm);
				for (size_t i : tally.flagged) {
					n_flagged++;
					fprintf(stderr, "Warning: read %s: device status %d (bit 0 = a work buffer overflowed), capacity site %u\n", b->reads[i].name.str().c_str(), (int)b->result(i).status, (unsigned)(b->result(i).pad2 & 0xffffu));
				}
				if (metrics) for (size_t i = 0; i < b->reads.size(); i++) {
					const ReadResult& rr = b->result(i);
					fprintf(stderr, "MET\t%s\titers=%u dps=%u ugs=%u bwseed=%u bwext=%u red=%u bt=%u nalns=%u extl=%u extr=%u res=%u\n", b->reads[i].name.str().c_str(),
					        rr.n_ex_iters, rr.n_ex_dps, rr.n_ex_ugs, rr.n_bwops_seed, rr.n_bwops_ext, rr.n_redundants, rr.n_bt_attempts, rr.nalns, rr.n_ext_left, rr.n_ext_right, rr.n_resolve_steps);
				}
				const auto tf1_ = std::chrono::steady_clock::now();
				uint64_t nbytes = 0;
				for (const std::string& part : parts) { put(part.data(), part.size()); nbytes += part.size(); }
				if (shard_idx && !b->reads.empty()) fprintf(shard_idx, "B %llu %llu %llu\n", (unsigned long long)b->block_id, (unsigned long long)nbytes, (unsigned long long)b->reads.size());
				t_format += std::chrono::duration<double>(tf1_ - tf0_).count();
				t_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf1_).count();
				const bool was_last = b->last;
				pool.put(std::move(b));
				if (was_last) { done = true; break; }
			}
		}
	});

	// Device stage: kWorkersPerDev threads per device, each with its own stream and buffers, so that one batch's
	// upload and another's download overlap the alignment of a third -- and so do the alignments themselves: the context
	// keeps one working set per stream (bt2g_align_batch), the kernels of the batches in flight share the device, and the
	// tail of one batch's worker kernel is filled by the next batch's waves.  The writer puts the batches back in input order.
	PinnedPool pinned;
	auto device_worker = [&](size_t d) {
		HIP_OK(hipSetDevice(devices[d]));
		bt2g_ctx* ctx = ctxs[d];
		hipStream_t st;
		HIP_OK(hipStreamCreate(&st));
		DevBuf d_seq, d_qual, d_off, d_rp, d_res, d_packed, d_poff;
		for (;;) {
			BatchPtr b = q_in.pop();
			if (b->terminator) break;
			if (!b->bad_input.empty()) die(b->bad_input);
			if (!b->too_long.empty()) die("read " + b->too_long + " is longer than " + std::to_string(BT2G_MAX_READ_LEN) + " bp (not supported on the device path)");
			// (said here, by name, rather than as bt2g_align_batch's refusal of the batch: ADVICE r5)
			if (P.khits > 64 && b->max_len > 512) die("-k above 64 (and -a) is not supported for reads longer than 512 bp by this build; this input has a read of " + std::to_string(b->max_len) + " bp");
			const size_t n = b->reads.size();
			b->res_off.assign(n + 1, 0);
			if (n > 0) {
				d_seq.ensure(b->seq.size() + 16); d_qual.ensure(b->qual.size() + 16);
				d_off.ensure(b->off.size() * 8); d_rp.ensure(n * sizeof(ReadParams));
				d_res.ensure(n * stride); d_packed.ensure(n * stride); d_poff.ensure((n + 1) * 8);
				const auto th0 = std::chrono::steady_clock::now();
				HIP_OK(hipMemcpyAsync(d_seq.p, b->seq.data(), b->seq.size(), hipMemcpyHostToDevice, st));
				HIP_OK(hipMemcpyAsync(d_qual.p, b->qual.data(), b->qual.size(), hipMemcpyHostToDevice, st));
				HIP_OK(hipMemcpyAsync(d_off.p, b->off.data(), b->off.size() * 8, hipMemcpyHostToDevice, st));
				HIP_OK(hipMemcpyAsync(d_rp.p, b->rp.data(), n * sizeof(ReadParams), hipMemcpyHostToDevice, st));
				bt2g_reads rd;
				rd.d_seq = (const uint8_t*)d_seq.p; rd.d_qual = (const uint8_t*)d_qual.p; rd.d_off = (const uint64_t*)d_off.p; rd.n_reads = (uint32_t)n;
				HIP_OK(hipStreamSynchronize(st));
				{
					auto ta = std::chrono::steady_clock::now();
					{ std::lock_guard<std::mutex> g2(align_mu); t_h2d += std::chrono::duration<double>(ta - th0).count(); }
					// the host derived every read's seed parameters, so it knows the widest seed table of the batch: with the bound in
					// the parameters bt2g_align_batch does not have to wait for the device to count them
					AlignParams Pb = P;
					Pb.paired = b->paired ? 1 : 0;       // (differs from P.paired only for the unpaired batches of a mixed run)
					uint32_t max_seeds = 1;
					for (size_t i = 0; i < n; i++) {
						const uint32_t len = (uint32_t)(b->off[i + 1] - b->off[i]);
						const uint32_t L = (uint32_t)b->rp[i].seedlen, iv = (uint32_t)(b->rp[i].interval > 0 ? b->rp[i].interval : 1);
						const uint32_t ns = 1 + (len > L ? (len - L) / iv : 0u);
						if (ns > max_seeds) max_seeds = ns;
					}
					const uint32_t offs_cap = b->max_len > 512 ? 128u : 64u;      // kMaxOffs of the worker class the batch will run in: the worker flags reads beyond it
					Pb.max_seeds = (int32_t)(max_seeds > offs_cap ? offs_cap : max_seeds);
					Pb.max_dp_cols = 0;
					if (b->paired) {
						// the widest window in which any mate of this batch is looked for next to its partner: beyond the default the launch
						// holds more per-column state (-X 800, --local pairs of long reads)
						uint32_t wmax = 0;
						for (size_t i = 0; i + 1 < n; i += 2) {
							const uint32_t l1 = (uint32_t)(b->off[i + 1] - b->off[i]), l2 = (uint32_t)(b->off[i + 2] - b->off[i + 1]);
							const uint32_t w1 = mate_window_bound(Pb, b->rp[i].minsc, l1, l1, l2), w2 = mate_window_bound(Pb, b->rp[i + 1].minsc, l2, l1, l2);
							if (w1 > wmax) wmax = w1;
							if (w2 > wmax) wmax = w2;
						}
						Pb.max_dp_cols = (int32_t)(wmax > (uint32_t)BT2G_MAX_DP_COLS ? (uint32_t)BT2G_MAX_DP_COLS : wmax);
					}
					int rc = bt2g_align_batch(ctx, &rd, (const bt2g_read_params*)d_rp.p, &Pb, b->max_len, d_res.p, st);
					if (rc) die(std::string("bt2g_align_batch: ") + bt2g_last_error(ctx));
					rc = bt2g_results_pack(ctx, d_res.p, (uint32_t)n, (uint32_t)P.khits, d_packed.p, (uint64_t*)d_poff.p, st);
					if (rc) die(std::string("bt2g_results_pack: ") + bt2g_last_error(ctx));
					HIP_OK(hipStreamSynchronize(st));
					std::lock_guard<std::mutex> g2(align_mu);
					align_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
				}
				const auto td0 = std::chrono::steady_clock::now();
				HIP_OK(hipMemcpyAsync(b->res_off.data(), d_poff.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
				HIP_OK(hipStreamSynchronize(st));
				const size_t total = (size_t)b->res_off[n];
				b->res_hold = pinned.get(total);
				b->res = (const uint8_t*)b->res_hold.get();
				HIP_OK(hipMemcpyAsync(b->res_hold.get(), d_packed.p, total, hipMemcpyDeviceToHost, st));
				HIP_OK(hipStreamSynchronize(st));
				{ std::lock_guard<std::mutex> g2(align_mu); t_d2h += std::chrono::duration<double>(std::chrono::steady_clock::now() - td0).count();
				  const double ts = since_load(); if (ts_first_aligned < 0) ts_first_aligned = ts; ts_last_aligned = ts; }
			}
			q_out.push(std::move(b));
		}
		(void)hipStreamDestroy(st);
	};
	{
		std::vector<std::thread> workers;
		for (size_t d = 0; d < ndev; d++) for (size_t w = 0; w < kWorkersPerDev; w++) workers.emplace_back(device_worker, d);
		for (auto& t : workers) t.join();
	}
	reader.join();
	writer.join();
	cleanup.out = nullptr;
	if (out != stdout ? fclose(out) != 0 : fflush(out) != 0) die(std::string("could not write SAM output: ") + strerror(errno));
	if (opt.timing) {
		auto hms = [](double s) { char b[64]; int h = (int)(s / 3600); int m = (int)(s / 60) % 60; int sec = (int)s % 60; snprintf(b, sizeof b, "%02d:%02d:%02d", h, m, sec); return std::string(b); };
		fprintf(stderr, "Time loading forward index: %s\n", hms(std::chrono::duration<double>(t1 - t0).count()).c_str());
		const double search_wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
		fprintf(stderr, "Multiseed full-index search: %s\n", hms(search_wall).c_str());
		fprintf(stderr, "[bt2g] index load %.3f s; search %.3f s wall, %llu reads -> %.0f reads/s after the load\n", std::chrono::duration<double>(t1 - t0).count(), search_wall,
		        (unsigned long long)(summ.nread + 2 * psumm.npair), search_wall > 0 ? (double)(summ.nread + 2 * psumm.npair) / search_wall : 0.0);
		fprintf(stderr, "[bt2g] after the load: first batch parsed at %.3f s, first batch back from the device at %.3f s, last batch parsed at %.3f s, last batch back at %.3f s, output closed at %.3f s\n",
		        ts_first_parsed, ts_first_aligned, ts_last_parsed, ts_last_aligned, search_wall);
		fprintf(stderr, "[bt2g] device stage, summed over its %zu threads: upload %.3f s, kernels + pack %.3f s, download %.3f s\n", ndev * kWorkersPerDev, t_h2d, align_s, t_d2h);
		fprintf(stderr, "[bt2g] host stages: split %.3f s, parse %.3f s, pack %.3f s, format %.3f s, write %.3f s\n", fq.t_split, fq.t_parse, fq.t_pack, t_format, t_write);
	}
	if (shard_idx) {
		// summary counters of this shard, for the cross-rank sum (the reference merges per-thread ReportingMetrics the same way, aln_sink.cpp:33-101)
		fprintf(shard_idx, "S %llu %llu %llu %llu\n", (unsigned long long)summ.nread, (unsigned long long)summ.n0, (unsigned long long)summ.nuni, (unsigned long long)summ.nrep);
		fprintf(shard_idx, "P %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", (unsigned long long)psumm.npair, (unsigned long long)psumm.conc0, (unsigned long long)psumm.conc_uni1,
		        (unsigned long long)psumm.conc_uni2, (unsigned long long)psumm.conc_rep, (unsigned long long)psumm.ndiscord, (unsigned long long)psumm.unp00,
		        (unsigned long long)psumm.unp0_uni1, (unsigned long long)psumm.unp0_uni2, (unsigned long long)psumm.unp0_rep);
		fprintf(shard_idx, "F %llu\n", n_flagged);
		fprintf(shard_idx, "R %llu\n", (unsigned long long)(fq.bytes_read() + (fq2 ? fq2->bytes_read() : 0)));      // bytes of reads files this rank took in
		cleanup.idx = nullptr;
		const bool idx_bad = ferror(shard_idx) != 0;
		if (fclose(shard_idx) != 0 || idx_bad) die("could not write " + ex.shard_index);
	}
	if (!opt.quiet && ex.shard_world == 1 && !ex.shard_bytes) { if (opt.mixed_unpaired) print_mixed_summary(stderr, psumm, summ, !opt.no_discordant, !opt.no_mixed); else if (opt.paired) psumm.print(stderr, !opt.no_discordant, !opt.no_mixed); else summ.print(stderr); }   // gQuiet (bt2_search.cpp:5198); sharded: rank 0 of the driver prints the merged summary
	for (bt2g_ctx*& c : ctxs) { bt2g_ctx_destroy(c); c = nullptr; }
	if (n_flagged) {
		// never pass off a capacity-limited result as the reference's
		fprintf(stderr, "Error: %llu read(s) exceeded a limit of this build (see the warnings above); their SAM records may differ from bowtie2's\n", (unsigned long long)n_flagged);
		return 1;
	}
	return 0;
}

extern "C" int bowtie(int argc, const char** argv) {
	g_caller = std::this_thread::get_id();
	try { return run_search(argc, const_cast<char**>(argv)); }
	catch (const CliExit& e) { return e.code; }
	catch (const std::exception& e) { fprintf(stderr, "Error: %s\n", e.what()); return 1; }
}

// The option -> parameter mapping of the executable for a caller that feeds bt2g_align_batch itself (bench.py's timed batches, the
// tests): argv as bowtie2-align-{s,l} takes it (index / read / output files may be left out), `read_len` the length of the reads the
// batch will hold.  *P = the batch parameters Options::to_params derives, *rp = the per-read parameters of an N-free read of that length
// (minimum score, seed interval -- times 1.2 when `both_mates_pass` --, N ceiling, seed length, all filters passed; the per-read RNG seed
// is the caller's to fill in, gen_rand_seed needs the read).  Returns 0, or 1 with the parser's message on stderr.
extern "C" int bt2g_cli_params(int argc, const char** argv, uint32_t read_len, int large_index, int both_mates_pass, bt2g_align_params* P, bt2g_read_params* rp) {
	if (!P || !rp) return 1;
	Options opt;
	CliExtra ex;
	ex.allow_paired = true;
	std::vector<char*> av;
	static char prog[] = "bowtie2-align";
	av.push_back(prog);
	for (int i = 0; i < argc; i++) av.push_back(const_cast<char*>(argv[i]));
	const std::string err = parse_cli((int)av.size(), av.data(), opt, ex);
	if (!err.empty()) { fprintf(stderr, "Error: %s\n", err.c_str()); return 1; }
	opt.to_params(*P, large_index != 0);
	const std::string bases(read_len, (char)0), quals(read_len, 'I');
	ReadRec r;
	r.seq.set(bases.data(), bases.size()); r.qual.set(quals.data(), quals.size());
	*rp = compute_read_params(opt, r);
	rp->seed = 0;
	if (both_mates_pass) { int iv = (int)(rp->interval * 1.2 + 0.5); rp->interval = iv < 1 ? 1 : iv; }
	return 0;
}
