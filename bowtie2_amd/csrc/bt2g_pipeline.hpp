// bt2g_pipeline.hpp -- host side of the drop-in binary: read ingest and SAM output as a three-stage
// pipeline around the device batch.
//
//   stage 1  reader thread   gz/plain FASTQ -> raw records (memchr line splitting), then a parallel pass that
//                            builds ReadRec + per-read parameters and packs the device arrays
//   stage 2  caller's thread H2D, bt2g_align_batch, D2H
//   stage 3  writer thread   parallel SAM formatting per chunk, ordered write, alignment summary
//
// Batches flow through bounded queues, so parsing batch i+1 and formatting batch i-1 overlap the device
// work on batch i.  Output order is input order regardless of -p (always a valid answer to --reorder).
#ifndef BT2G_PIPELINE_HPP_
#define BT2G_PIPELINE_HPP_

#include <atomic>
#include <zlib.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>

#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bt2g_host.hpp"

namespace bt2g {

// cores this process may actually use: min(affinity mask, cgroup cpu.max quota)
inline unsigned usable_cores() {
	unsigned n = std::thread::hardware_concurrency();
	if (n == 0) n = 1;
	std::ifstream f("/sys/fs/cgroup/cpu.max");
	std::string quota, period;
	if (f >> quota >> period) {
		if (quota != "max") {
			const double q = atof(quota.c_str()), p = atof(period.c_str());
			if (q > 0 && p > 0) { const unsigned c = (unsigned)(q / p + 0.5); if (c >= 1 && c < n) n = c; }
		}
	}
	return n;
}

// run fn(chunk_index) for chunk_index in [0, nchunks) on up to `threads` threads
inline void parallel_for(size_t nchunks, unsigned threads, const std::function<void(size_t)>& fn) {
	if (threads <= 1 || nchunks <= 1) { for (size_t i = 0; i < nchunks; i++) fn(i); return; }
	std::mutex m;
	size_t next = 0;
	auto body = [&]() {
		for (;;) {
			size_t i;
			{ std::lock_guard<std::mutex> g(m); if (next >= nchunks) return; i = next++; }
			fn(i);
		}
	};
	std::vector<std::thread> pool;
	const unsigned nt = (unsigned)std::min<size_t>(threads, nchunks);
	for (unsigned t = 1; t < nt; t++) pool.emplace_back(body);
	body();
	for (auto& t : pool) t.join();
}

template <typename T>
class BoundedQueue {
public:
	explicit BoundedQueue(size_t cap) : cap_(cap) {}
	void push(T v) {
		std::unique_lock<std::mutex> l(m_);
		not_full_.wait(l, [&] { return q_.size() < cap_; });
		q_.push_back(std::move(v));
		not_empty_.notify_one();
	}
	T pop() {
		std::unique_lock<std::mutex> l(m_);
		not_empty_.wait(l, [&] { return !q_.empty(); });
		T v = std::move(q_.front());
		q_.pop_front();
		not_full_.notify_one();
		return v;
	}
private:
	size_t cap_;
	std::mutex m_;
	std::condition_variable not_full_, not_empty_;
	std::deque<T> q_;
};

// One batch travelling through the pipeline
struct HostBatch {
	struct Chunk { std::string names, seq, qual, orig, tags, error; };   // arenas of 4096 reads: names stay here, seq/qual are copied to the packed arrays
	std::vector<Chunk> chunks;
	std::vector<ReadRec> reads;                       // views into `chunks` (names) and `seq`/`qual` (packed arrays)
	std::vector<uint8_t> seq, qual;       // packed device arrays
	std::vector<uint64_t> off;
	std::vector<ReadParams> rp;
	uint32_t max_len = 0;
	// packed result records (bt2g_results_pack), filled by the device stage: record i starts at res + res_off[i]
	const uint8_t* res = nullptr;
	std::vector<uint64_t> res_off;
	std::shared_ptr<void> res_hold;       // keeps the (pinned, pooled) buffer behind `res` until the batch is written
	const ReadResult& result(size_t i) const { return *(const ReadResult*)(res + res_off[i]); }
	static const AlnRes* next_aln(const AlnRes* a) {
		return (const AlnRes*)((const uint8_t*)a + ((offsetof(AlnRes, ned) + (size_t)a->nned * sizeof(Edit) + 7) & ~(size_t)7));
	}
	std::string too_long;                 // name of a read over the length limit (fatal), if any
	std::string bad_input;                // malformed or unsupported input record (fatal), if any
	bool last = false;                    // end-of-input marker (may still carry reads)
	bool upto_hit = false;                // ... and the end is -u's, not the file's: a mixed run then stops instead of going on to the unpaired files
	bool terminator = false;              // tells one device worker to stop (carries nothing)
	uint64_t seqno = 0;                   // position in the input, for ordered output with several devices
	uint64_t block_id = 0;                // --shard: which block of the input this batch is
	void clear_reads() { chunks.clear(); reads.clear(); seq.clear(); qual.clear(); off.assign(1, 0); rp.clear(); max_len = 0; too_long.clear(); mate_src[0].reset(); mate_src[1].reset(); }
	// Back to the state of a new batch, but with every buffer's memory kept: a batch of 262 144 reads touches about 250 MB, and memory that
	// comes fresh from the allocator is paid for in page faults -- about as long as the device needs for the batch.  The driver hands finished
	// batches back to the reader instead of freeing them (bt2g_search.cpp: BatchPool).
	void recycle() {
		for (Chunk& c : chunks) { c.names.clear(); c.seq.clear(); c.qual.clear(); c.orig.clear(); c.tags.clear(); c.error.clear(); }
		reads.clear(); seq.clear(); qual.clear(); off.clear(); rp.clear(); max_len = 0;
		res = nullptr; res_off.clear(); res_hold.reset();
		too_long.clear(); bad_input.clear();
		last = upto_hit = terminator = false; seqno = block_id = 0; paired = false;
	}
	// paired-end: reads[2i] / reads[2i+1] are the mates of pair i; their text lives in the two single-mate batches
	bool paired = false;
	std::unique_ptr<HostBatch> mate_src[2];
};

// Interleave two single-mate batches into one paired batch (mate 1 at even, mate 2 at odd positions), and apply the
// pair-level adjustments of the per-read parameters (bt2_search.cpp:3427-3434: seed interval x1.2 when both mates pass
// their filters).
inline void merge_mate_batches(std::unique_ptr<HostBatch> b1, std::unique_ptr<HostBatch> b2, HostBatch& out, const Options& opt) {
	out.paired = true;
	out.last = b1->last || b2->last;
	out.upto_hit = b1->upto_hit || b2->upto_hit;
	out.bad_input = !b1->bad_input.empty() ? b1->bad_input : b2->bad_input;
	out.too_long = !b1->too_long.empty() ? b1->too_long : b2->too_long;
	if (out.bad_input.empty() && b1->reads.size() != b2->reads.size())
		out.bad_input = "fewer reads in file specified with -" + std::string(b1->reads.size() < b2->reads.size() ? "1" : "2") + " than in file specified with -" + (b1->reads.size() < b2->reads.size() ? "2" : "1");
	const size_t n = std::min(b1->reads.size(), b2->reads.size());
	out.reads.resize(2 * n); out.rp.resize(2 * n); out.off.resize(2 * n + 1);
	out.off[0] = 0;
	out.max_len = std::max(b1->max_len, b2->max_len);
	for (size_t i = 0; i < n; i++) {
		out.reads[2 * i] = b1->reads[i]; out.reads[2 * i + 1] = b2->reads[i];
		out.rp[2 * i] = b1->rp[i]; out.rp[2 * i + 1] = b2->rp[i];
		if ((out.rp[2 * i].filt & 15u) == 15u && (out.rp[2 * i + 1].filt & 15u) == 15u) {
			for (int m = 0; m < 2; m++) {
				int iv = opt.ms_ival.f<int>((double)out.reads[2 * i + m].seq.size());
				iv = (int)(iv * 1.2 + 0.5);
				out.rp[2 * i + m].interval = iv < 1 ? 1 : iv;
			}
		}
		out.off[2 * i + 1] = out.off[2 * i] + b1->reads[i].seq.size();
		out.off[2 * i + 2] = out.off[2 * i + 1] + b2->reads[i].seq.size();
	}
	out.seq.resize(out.off[2 * n]); out.qual.resize(out.off[2 * n]);
	for (size_t i = 0; i < 2 * n; i++) {
		const ReadRec& r = out.reads[i];
		if (r.seq.size()) { memcpy(&out.seq[out.off[i]], r.seq.data(), r.seq.size()); memcpy(&out.qual[out.off[i]], r.qual.data(), r.qual.size()); }
	}
	out.mate_src[0] = std::move(b1); out.mate_src[1] = std::move(b2);
}

// Line-oriented reader over gz or plain input (gzread handles both)
class LineSource {
public:
	// `path` may be a comma-separated list of files (as -U / -1 / -2 accept, bt2_search.cpp:1185-1199); they are read one after the other
	explicit LineSource(const std::string& path) {
		for (size_t a = 0; a <= path.size();) { size_t b = path.find(',', a); if (b == std::string::npos) b = path.size(); if (b > a) paths_.push_back(path.substr(a, b - a)); a = b + 1; }
		if (paths_.empty()) paths_.push_back(path);
		open_next();
	}
	~LineSource() { if (f_) gzclose(f_); if (plain_fd_ >= 0) close(plain_fd_); }
	bool ok() const { return f_ != nullptr || plain_fd_ >= 0; }
	// Restrict the source to bytes [a, b) of its (single, uncompressed, seekable) file: what one rank of a byte-sharded run reads
	// (--shard-bytes).  The caller vouches that `a` is the start of a line and that the line ending at b - 1 ends a record.
	bool set_range(uint64_t a, uint64_t b, std::string& err) {
		if (paths_.size() != 1 || paths_[0] == "-" || !ok()) { err = "--shard-bytes needs one regular reads file per -U / -1 / -2"; return false; }
		if (plain_fd_ < 0 && !gzdirect(f_)) { err = "--shard-bytes cannot be used with compressed input (" + paths_[0] + ")"; return false; }
		if (b < a || (plain_fd_ >= 0 ? lseek(plain_fd_, (off_t)a, SEEK_SET) < 0 : gzseek(f_, (z_off_t)a, SEEK_SET) < 0)) { err = "--shard-bytes: cannot seek in " + paths_[0]; return false; }
		ranged_ = true; left_ = b - a; bytes_read_.store(0, std::memory_order_relaxed);
		return true;
	}
	uint64_t bytes_read() const { return bytes_read_.load(std::memory_order_relaxed); }      // bytes taken from the file(s) so far (read by the driver while the prefetch thread refills)
	uint64_t plain_size() const {
		if (paths_.size() != 1 || paths_[0] == "-" || !ok() || (plain_fd_ < 0 && !gzdirect(f_))) return 0;
		struct stat st_;
		if (stat(paths_[0].c_str(), &st_) != 0 || !S_ISREG(st_.st_mode)) return 0;
		return ranged_ ? 0 : (uint64_t)st_.st_size;
	}
	// next line without its terminator ('\n' or "\r\n"); false at end of input.  The view is valid until the next call.
	bool next(const char*& p, size_t& n) {
		for (;;) {
			const char* base = buf_.get() + pos_;
			const size_t avail = len_ - pos_;
			const char* nl = avail ? (const char*)memchr(base, '\n', avail) : nullptr;
			if (nl) {
				p = base; n = (size_t)(nl - base);
				pos_ += n + 1;
				raw_len_ = n + 1;
				if (n && p[n - 1] == '\r') n--;
				return true;
			}
			if (eof_) {
				if (avail == 0) return false;
				p = base; n = avail; pos_ = len_;
				raw_len_ = n;
				unterminated_ = true;          // last line of the input has no newline
				if (n && p[n - 1] == '\r') n--;
				return true;
			}
			refill();
		}
	}
	// One whole 4-line FASTQ record if all of it is in the buffer: rec -> its first byte, e[k] = offset of the k-th line's '\n' from rec.  False: fewer than
	// four line ends are buffered (the caller falls back to next(), which refills), or a line ends in "\r\n" (left to next()).  The view is valid
	// until the next call; nothing is consumed.  (The serial scan of the reader: one pass and one copy per record instead of four calls and three copies.)
	bool peek_record4(const char*& rec, uint32_t (&e)[4], size_t& nbytes) const {
		const char* base = buf_.get() + pos_;
		size_t avail = len_ - pos_, o = 0;
		for (int k = 0; k < 4; k++) {
			const char* nl = avail > o ? (const char*)memchr(base + o, '\n', avail - o) : nullptr;
			if (!nl) return false;
			e[k] = (uint32_t)(nl - base);
			if (e[k] > 0 && nl[-1] == '\r') return false;
			o = (size_t)(nl - base) + 1;
		}
		rec = base; nbytes = o;
		return true;
	}
	void consume(size_t nbytes) { pos_ += nbytes; raw_len_ = nbytes; }      // ... and take it
private:
	// (the buffer is plain storage with its own length: a std::string would zero-fill the 8 MB of every refill before gzread overwrites them)
	void reserve(size_t need) {
		if (need <= cap_) return;
		size_t nc = cap_ ? cap_ : (size_t)(16u << 20);
		while (nc < need) nc *= 2;
		std::unique_ptr<char[]> nb(new char[nc]);
		if (len_) memcpy(nb.get(), buf_.get(), len_);
		buf_ = std::move(nb); cap_ = nc;
	}
	void refill() {
		if (pos_ > 0) { if (len_ > pos_) memmove(buf_.get(), buf_.get() + pos_, len_ - pos_); len_ -= pos_; pos_ = 0; }
		const size_t old = len_;
		size_t want = 8u << 20;
		if (ranged_ && (uint64_t)want > left_) want = (size_t)left_;
		reserve(old + want + 1);
		// (a plain file is read with read(2): gzread's transparent mode copies everything once more through its own buffer)
		int got = 0;
		if (want) {
			if (plain_fd_ >= 0) { ssize_t r_; do { r_ = ::read(plain_fd_, buf_.get() + old, want); } while (r_ < 0 && errno == EINTR); got = (int)r_; }
			else got = gzread(f_, buf_.get() + old, (unsigned)want);
		}
		len_ = old + (got > 0 ? (size_t)got : 0);
		if (got > 0) { bytes_read_.fetch_add((uint64_t)got, std::memory_order_relaxed); if (ranged_) left_ -= (uint64_t)got; }
		if (got < 0) io_error_ = true;          // corrupt / truncated .gz: the run must fail, not end early (the reference aborts too)
		if (got <= 0) {
			if (next_path_ < paths_.size()) {
				// next file of the list: the previous one ends a line even if its last newline is missing
				if (len_ && buf_[len_ - 1] != '\n') { reserve(len_ + 1); buf_[len_++] = '\n'; }
				if (f_) gzclose(f_);
				f_ = nullptr;
				if (plain_fd_ >= 0) { close(plain_fd_); plain_fd_ = -1; }
				if (!open_next()) eof_ = true;
			} else eof_ = true;
		}
	}
	bool open_next() {
		const std::string& p = paths_[next_path_++];
		if (p != "-") {
			// a regular file that does not start with the gzip magic is read directly
			const int fd = open(p.c_str(), O_RDONLY | O_CLOEXEC);
			if (fd >= 0) {
				struct stat st_;
				unsigned char m[2] = {0, 0};
				if (fstat(fd, &st_) == 0 && S_ISREG(st_.st_mode) && (pread(fd, m, 2, 0) < 2 || !(m[0] == 0x1f && m[1] == 0x8b))) { plain_fd_ = fd; return true; }
				close(fd);
			}
		}
		f_ = p == "-" ? gzdopen(0, "rb") : gzopen(p.c_str(), "rb");
		if (f_) gzbuffer(f_, 1 << 20);
		return f_ != nullptr;
	}
	std::vector<std::string> paths_;
	size_t next_path_ = 0;
	bool unterminated_ = false;
	bool io_error_ = false;
	bool ranged_ = false;
	uint64_t left_ = 0;
	std::atomic<uint64_t> bytes_read_{0};
public:
	bool last_line_unterminated() const { return unterminated_; }
	bool io_error() const { return io_error_; }
	size_t last_raw_len() const { return raw_len_; }      // length of the last line including its terminator (--passthrough)
private:
	size_t raw_len_ = 0;
	gzFile f_ = nullptr;
	int plain_fd_ = -1;
	std::unique_ptr<char[]> buf_;
	size_t cap_ = 0, len_ = 0;
	size_t pos_ = 0;
	bool eof_ = false;
};

// read character -> code, 255 = not part of the sequence (see FastqBatcher::finish)
struct SeqCodeTable {
	unsigned char t[256];
	SeqCodeTable() { for (int c = 0; c < 256; c++) t[c] = (c == '.' || isalpha(c)) ? (unsigned char)asc2code(c == '.' ? 'N' : c) : (unsigned char)255; }
};
static const SeqCodeTable kSeqCode;

// Byte stream over a list of BAM files.  BAM is a series of BGZF blocks = concatenated gzip members, which gzread() inflates one after
// the other (the reference walks the blocks itself, BAMPatternSource::nextBGZFBlockFromFile, pat.cpp:1270-1323; the empty end-of-file
// block inflates to nothing either way).
class BamStream {
public:
	explicit BamStream(const std::string& path) {
		for (size_t a = 0; a <= path.size();) { size_t b = path.find(',', a); if (b == std::string::npos) b = path.size(); if (b > a) paths_.push_back(path.substr(a, b - a)); a = b + 1; }
		// Of a comma-separated list the reference reads the first file only: BAMPatternSource::nextBatch (pat.cpp:1325-1360) reports the end
		// of the input at the first file's last block and never opens the next one.  Same here, so that the output is the same.
		if (paths_.size() > 1) paths_.resize(1);
	}
	~BamStream() { if (f_) gzclose(f_); }
	// positions the stream on the first alignment record of the next file that has one; false when the list is exhausted
	bool next_file(std::string& err) {
		if (f_) { gzclose(f_); f_ = nullptr; }
		if (next_ >= paths_.size()) return false;
		const std::string& p = paths_[next_++];
		f_ = p == "-" ? gzdopen(0, "rb") : gzopen(p.c_str(), "rb");
		if (!f_) { err = "cannot open reads file " + p; return false; }
		gzbuffer(f_, 1 << 20);
		// magic, header text, reference dictionary (get_alignments, pat.cpp:1366-1385)
		char magic[4]; uint32_t l_text = 0, nref = 0;
		std::string skip;
		if (!read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { err = "reads file " + p + " is not a BAM file"; return false; }
		if (!read(&l_text, 4) || !read_skip(l_text) || !read(&nref, 4)) { err = "truncated BAM header in " + p; return false; }
		for (uint32_t j = 0; j < nref; j++) {
			uint32_t l_name = 0;
			if (!read(&l_name, 4) || !read_skip((size_t)l_name + 4)) { err = "truncated BAM header in " + p; return false; }
		}
		return true;
	}
	bool open() const { return f_ != nullptr; }
	// exactly n bytes, or false (clean end of file when nothing was left; `partial` tells a record cut short)
	bool read(void* dst, size_t n) {
		partial_ = false;
		size_t got = 0;
		while (got < n) {
			const int r = gzread(f_, (char*)dst + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
			if (r < 0) { io_error_ = true; return false; }
			if (r == 0) { partial_ = got > 0; return false; }
			got += (size_t)r;
		}
		return true;
	}
	bool partial() const { return partial_; }
	bool io_error() const { return io_error_; }
private:
	bool read_skip(size_t n) { tmp_.resize(n); return n == 0 || read(&tmp_[0], n); }
	std::vector<std::string> paths_;
	size_t next_ = 0;
	gzFile f_ = nullptr;
	bool partial_ = false, io_error_ = false;
	std::string tmp_;
};

// FASTQ records following FastqPatternSource::parse (pat.cpp): 4-line records, '.' -> N, non-letters dropped
class FastqBatcher {
public:
	FastqBatcher(const std::string& path, const Options& opt, unsigned threads) : src_((opt.format == 3 || opt.format == 7) ? std::string("/dev/null") : path), cmd_(path), bam_(opt.format == 7 ? path : std::string()), opt_(opt), threads_(threads) {
		unit_ = (!opt.interleaved_file.empty() && path == opt.interleaved_file) ? 2 : 1;
		ushift_ = unit_ == 2 ? 1 : 0;
		if (opt.format == 7) bam_ok_ = bam_.next_file(bam_err_);
	}
	bool ok() const { return opt_.format == 7 ? (bam_ok_ || bam_err_.empty()) : src_.ok(); }
	// byte-sharded run: this source reads bytes [a, b) of its file, whose first record is read / pair number `first_read` of the input
	bool set_range(uint64_t a, uint64_t b, uint64_t first_read, std::string& err) {
		if (opt_.format != 0) { err = "--shard-bytes is for FASTQ input"; return false; }
		if (!src_.set_range(a, b, err)) return false;
		rdid_ = first_read * unit_;
		return true;
	}
	uint64_t bytes_read() const { return src_.bytes_read(); }
	// a split() of the next batch may still be running on the helper thread: it reads src_ and the other members, so it must have ended
	// before any of them is destroyed (prefetch_ is declared before them and would otherwise be waited for last)
	~FastqBatcher() { if (prefetch_.valid()) prefetch_.wait(); }
	// -b with -1/-2: which mate's records this source keeps (--align-paired-reads: flag 0x40 for mate 1, 0x80 for mate 2, pat.cpp:1422-1427)
	void set_bam_mate(int m) { bam_mate_ = m; }
	std::string open_error(const std::string& dflt) const { return bam_err_.empty() ? dflt : bam_err_; }

	double t_split = 0, t_parse = 0, t_pack = 0;      // seconds spent in each part of next() (-t)
	std::vector<LenParams> len_params_;                // per read length (filled by the first finish())
	// Fill `b` with up to max_reads reads; sets b.last at end of input (or at -u).
	// Two stages.  split(): the serial scan of the input into records (one thread: it is a scan of a byte stream).  finish(): records ->
	// codes, qualities, names and per-read parameters, parallel over chunks.  While the caller finishes batch k, a helper thread already
	// splits batch k + 1 into the other RawBatch -- the serial scan no longer adds to the time of a batch, it only has to keep up.
	// next_max_reads: size of the batch after this one (its scan starts now), when the caller varies the batch size; 0 = the same.
	void next(HostBatch& b, size_t max_reads, size_t max_read_len, size_t next_max_reads = 0) {
		RawBatch* cur;
		if (prefetch_.valid()) { prefetch_.get(); cur = &raw_[cur_ ^= 1]; }
		else { cur = &raw_[cur_]; split(*cur, max_reads); }
		if (!cur->last && cur->bad_input.empty()) {
			RawBatch* nxt = &raw_[cur_ ^ 1];
			const size_t nmax = next_max_reads ? next_max_reads : max_reads;
			prefetch_ = std::async(std::launch::async, [this, nxt, nmax]() { split(*nxt, nmax); });
		}
		finish(*cur, b, max_read_len);
	}
	// size in bytes of the input when it is one regular uncompressed file (what is left of it can then be estimated), else 0
	uint64_t plain_size() const { return src_.plain_size(); }
private:
	struct Raw { uint64_t rdid; size_t name_off, name_len, seq_off, seq_len, qual_off, qual_len; bool has_qual; char filter; size_t orig_off, orig_len; size_t tag_off = 0, tag_len = 0; };
	// what split() hands to finish(): the records of one batch as offsets into one arena, and how the scan ended
	struct RawBatch { std::string arena, orig; std::vector<Raw> recs; bool last = false, upto_hit = false, io_error = false; std::string bad_input; };
	RawBatch raw_[2];
	int cur_ = 0;
	std::future<void> prefetch_;
	static double tnow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	void split(RawBatch& rb, size_t max_reads) {
		const double t0_ = tnow();
		// ---- serial part: split the text into records (line copies into one arena)
		std::string& arena_ = rb.arena; std::vector<Raw>& recs_ = rb.recs; std::string& orig_ = rb.orig;
		RawBatch& b = rb;       // (the scan's verdicts -- last, upto_hit, bad_input -- travel with the records)
		arena_.clear(); recs_.clear(); orig_.clear();
		b.last = b.upto_hit = b.io_error = false; b.bad_input.clear();
		const bool pt = opt_.passthrough;
		while (recs_.size() < max_reads) {
			const char* p; size_t n;
			Raw r, r2;
			bool have_r2 = false;
			r.qual_off = r.qual_len = 0; r.has_qual = false; r.filter = '1';
			r.orig_off = orig_.size(); r.orig_len = 0;
			if (opt_.format == 0 && !pt && fastq_started_) {
				// a whole record in the buffer (the common case): name, sequence and qualities in one copy
				const char* rec; uint32_t e[4]; size_t nb;
				if (src_.peek_record4(rec, e, nb) && e[0] > 0 && rec[0] == '@' && e[1] > e[0] + 1) {
					if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
					src_.consume(nb);
					const size_t a0 = arena_.size();
					arena_.append(rec + 1, e[3] - 1);
					r.name_off = a0; r.name_len = e[0] - 1;
					r.seq_off = a0 + e[0]; r.seq_len = e[1] - e[0] - 1;
					r.qual_off = a0 + e[2]; r.qual_len = e[3] - e[2] - 1; r.has_qual = true;
					goto have_record;
				}
				// (a record the fast path does not take -- blank line, empty sequence, no '@', "\r\n", the end of the buffer: the line-by-line path decides)
			}
			if (opt_.format == 0) {                    // FASTQ: 4-line records
				bool got;
				size_t nblank = 0;
				do { got = src_.next(p, n); if (got && n == 0) nblank++; } while (got && n == 0);        // blank lines between records
				// a file of nothing but blank lines is not an empty FASTQ file: the reference looks for '@' after them and aborts (pat.cpp:1070)
				if (!got) { if (!fastq_started_ && nblank > 0) b.bad_input = "reads file does not look like a FASTQ file"; b.last = true; break; }
				fastq_started_ = true;
				if (p[0] != '@') { b.bad_input = "reads file does not look like a FASTQ file"; b.last = true; break; }   // pat.cpp:1070
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				r.name_off = arena_.size(); r.name_len = n - 1; arena_.append(p + 1, n - 1);
				if (pt) orig_.append(p, src_.last_raw_len());
				// a record cut short by the end of the file is malformed input (the reference aborts: pat.cpp:1100-1180), not the end of the run
				const char* trunc = "reads file ends in the middle of a FASTQ record";
				if (!src_.next(p, n)) { b.bad_input = trunc; b.last = true; break; }
				r.seq_off = arena_.size(); r.seq_len = n; arena_.append(p, n);
				if (pt) orig_.append(p, src_.last_raw_len());
				if (!src_.next(p, n)) { b.bad_input = trunc; b.last = true; break; }            // '+' line
				if (pt) orig_.append(p, src_.last_raw_len());
				if (!src_.next(p, n)) { b.bad_input = trunc; b.last = true; break; }
				r.qual_off = arena_.size(); r.qual_len = n; arena_.append(p, n); r.has_qual = true;
				if (pt) orig_.append(p, src_.last_raw_len());
			} else if (opt_.format == 1) {             // FASTA: '>' name, sequence possibly over several lines
				bool got = true;
				if (!have_pending_) {
					size_t nblank = 0;
					do { got = src_.next(p, n); if (got && n == 0) nblank++; } while (got && n == 0);
					// nothing but blank lines: not an empty FASTA file for the reference, which looks for '>' after them (pat.cpp:785-796)
					if (!got && !fasta_started_ && nblank > 0) { b.bad_input = "reads file does not look like a FASTA file"; b.last = true; break; }
					fasta_started_ = true;
					if (got && p[0] != '>') { b.bad_input = "reads file does not look like a FASTA file"; b.last = true; break; }   // pat.cpp:794
					if (got) { pending_.assign(p, n); if (pt) pending_raw_.assign(p, src_.last_raw_len()); }
				}
				if (!got) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				have_pending_ = false;
				r.name_off = arena_.size(); r.name_len = pending_.size() - 1; arena_.append(pending_.data() + 1, pending_.size() - 1);
				r.seq_off = arena_.size(); r.seq_len = 0;
				if (pt) orig_ += pending_raw_;
				while (src_.next(p, n)) {
					if (n && p[0] == '>') { pending_.assign(p, n); have_pending_ = true; if (pt) pending_raw_.assign(p, src_.last_raw_len()); break; }
					if (pt) orig_.append(p, src_.last_raw_len());
					// the reference's FASTA parser drops the last character of a final line that lacks its newline
					if (src_.last_line_unterminated() && n > 0) n--;
					arena_.append(p, n); r.seq_len += n;
				}
			} else if (opt_.format == 3) {             // -c: reads given on the command line, "SEQ[:QUALS]" separated by commas
				if (cmd_pos_ > cmd_.size() || cmd_.empty()) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				size_t e = cmd_.find(',', cmd_pos_);
				if (e == std::string::npos) e = cmd_.size();
				const std::string tok = cmd_.substr(cmd_pos_, e - cmd_pos_);
				cmd_pos_ = e + 1;
				const size_t colon = tok.find(':');
				r.name_off = arena_.size(); r.name_len = 0;
				r.seq_off = arena_.size(); r.seq_len = colon == std::string::npos ? tok.size() : colon; arena_.append(tok.data(), r.seq_len);
				if (colon != std::string::npos) { r.qual_off = arena_.size(); r.qual_len = tok.size() - colon - 1; arena_.append(tok.data() + colon + 1, r.qual_len); r.has_qual = true; }
				if (pt) { orig_ += std::to_string((rdid_ >> ushift_)); orig_.push_back('\t'); orig_.append(tok.data(), r.seq_len); orig_.push_back('\t'); if (r.has_qual) orig_.append(tok.data() + colon + 1, r.qual_len); else orig_.append(r.seq_len, 'I'); }
			} else if (opt_.format == 4) {             // --tab5 / --tab6, unpaired form: name <tab> seq <tab> quals
				bool got;
				do { got = src_.next(p, n); } while (got && n == 0);
				if (!got) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				if (pt) orig_.append(p, n);
				const char* t1 = (const char*)memchr(p, '\t', n);
				const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(p + n - t1 - 1)) : nullptr;
				if (!t1 || !t2) { b.bad_input = "malformed tab-delimited read record"; b.last = true; break; }
				const char* t3 = (const char*)memchr(t2 + 1, '\t', (size_t)(p + n - t2 - 1));
				r.name_off = arena_.size(); r.name_len = (size_t)(t1 - p); arena_.append(p, r.name_len);
				r.seq_off = arena_.size(); r.seq_len = (size_t)(t2 - t1 - 1); arena_.append(t1 + 1, r.seq_len);
				if (!t3) {
					if (unit_ == 2) { b.bad_input = "unpaired record in a paired tab-delimited file (mixing is not supported by this build)"; b.last = true; break; }
					r.qual_off = arena_.size(); r.qual_len = (size_t)(p + n - t2 - 1); arena_.append(t2 + 1, r.qual_len); r.has_qual = true;
				} else {
					// paired record: name seq1 qual1 seq2 qual2 (--tab5) or name1 seq1 qual1 name2 seq2 qual2 (--tab6)
					// (TabbedPatternSource::parse, pat.cpp:1545-1660)
					if (unit_ != 2) { b.bad_input = "paired record in an unpaired tab-delimited file (mixing is not supported by this build)"; b.last = true; break; }
					r.qual_off = arena_.size(); r.qual_len = (size_t)(t3 - t2 - 1); arena_.append(t2 + 1, r.qual_len); r.has_qual = true;
					const char* f[4]; int nf = 0; f[nf++] = t3 + 1;
					for (const char* q = t3 + 1; q < p + n && nf < 4; q++) if (*q == '\t') f[nf++] = q + 1;
					const char* end = p + n;
					r2 = r; have_r2 = true;
					if (nf == 2) {            // seq2 qual2, name shared
						r2.name_off = r.name_off; r2.name_len = r.name_len;
						r2.seq_off = arena_.size(); r2.seq_len = (size_t)(f[1] - 1 - f[0]); arena_.append(f[0], r2.seq_len);
						r2.qual_off = arena_.size(); r2.qual_len = (size_t)(end - f[1]); arena_.append(f[1], r2.qual_len); r2.has_qual = true;
					} else if (nf == 3) {     // name2 seq2 qual2
						r2.name_off = arena_.size(); r2.name_len = (size_t)(f[1] - 1 - f[0]); arena_.append(f[0], r2.name_len);
						r2.seq_off = arena_.size(); r2.seq_len = (size_t)(f[2] - 1 - f[1]); arena_.append(f[1], r2.seq_len);
						r2.qual_off = arena_.size(); r2.qual_len = (size_t)(end - f[2]); arena_.append(f[2], r2.qual_len); r2.has_qual = true;
					} else { b.bad_input = "malformed paired tab-delimited read record"; b.last = true; break; }
				}
			} else if (opt_.format == 5) {             // --qseq: 11 tab-separated fields (read_qseq.cpp:83-233)
				bool got;
				do { got = src_.next(p, n); } while (got && n == 0);
				if (!got) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				const char* f[12]; int nf = 0; f[nf++] = p;
				for (size_t k = 0; k < n && nf < 12; k++) if (p[k] == '\t') f[nf++] = p + k + 1;
				if (nf != 11) { b.bad_input = "malformed QSEQ record (expected 11 fields)"; b.last = true; break; }
				f[11] = p + n + 1;
				auto fld = [&](int k) { return std::string(f[k], (size_t)(f[k + 1] - f[k] - 1)); };
				// name = machine_run_lane_tile_x_y_index/read
				std::string nm;
				for (int k = 0; k < 7; k++) { nm += fld(k); nm.push_back(k == 6 ? '/' : '_'); }
				nm += fld(7);
				const std::string sq = fld(8), ql = fld(9), fl = fld(10);
				if (fl != "0" && fl != "1") { b.bad_input = "Bad value '" + fl + "' for qseq filter flag"; b.last = true; break; }
				r.name_off = arena_.size(); r.name_len = nm.size(); arena_.append(nm);
				r.seq_off = arena_.size(); r.seq_len = sq.size(); arena_.append(sq);
				r.qual_off = arena_.size(); r.qual_len = ql.size(); arena_.append(ql); r.has_qual = true;
				r.filter = fl[0];
			} else if (opt_.format == 7) {             // -b: unaligned BAM (BAMPatternSource::get_alignments / parse, pat.cpp:1362-1515)
				bool got = false;
				while (bam_ok_ && !got) {
					uint32_t block_size = 0;
					if (!bam_.read(&block_size, 4)) {
						if (bam_.partial() || bam_.io_error()) { b.bad_input = "error while reading the BAM file (truncated or corrupt)"; break; }
						bam_ok_ = bam_.next_file(bam_err_);
						if (!bam_ok_ && !bam_err_.empty()) b.bad_input = bam_err_;
						continue;
					}
					if (block_size == 0) { bam_ok_ = bam_.next_file(bam_err_); if (!bam_ok_ && !bam_err_.empty()) b.bad_input = bam_err_; continue; }
					if (block_size > (1u << 28)) { b.bad_input = "error while reading the BAM file (implausible record length: corrupt input?)"; break; }
					bam_rec_.resize(block_size);
					if (block_size < 32 || !bam_.read(&bam_rec_[0], block_size)) { b.bad_input = "error while reading the BAM file (truncated or corrupt)"; break; }
					uint16_t flag; memcpy(&flag, bam_rec_.data() + 14, 2);
					// only unaligned records are reads; unpaired ones by default, with --align-paired-reads the paired ones of this source's mate
					if (!(flag & 0x4)) continue;
					if (!opt_.align_paired_reads && (flag & 0x1)) continue;
					if (opt_.align_paired_reads && bam_mate_ && !(flag & (bam_mate_ == 1 ? 0x40 : 0x80))) continue;
					const uint8_t l_read_name = (uint8_t)bam_rec_[8];
					uint16_t n_cigar; memcpy(&n_cigar, bam_rec_.data() + 12, 2);
					int32_t l_seq; memcpy(&l_seq, bam_rec_.data() + 16, 4);
					size_t off = 32;
					const size_t need = off + l_read_name + 4u * n_cigar + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
					if (l_seq < 0 || l_read_name == 0 || need > block_size) { b.bad_input = "malformed BAM record"; break; }
					r.name_off = arena_.size(); r.name_len = (size_t)l_read_name - 1; arena_.append(bam_rec_.data() + off, r.name_len);
					off += l_read_name + 4u * n_cigar;
					const unsigned char* sq = (const unsigned char*)bam_rec_.data() + off;
					off += (size_t)((l_seq + 1) / 2);
					r.seq_off = arena_.size(); r.seq_len = (size_t)l_seq;
					// 4-bit codes; everything but A/C/G/T becomes N downstream, as asc2dna does (pat.cpp:1494-1495): '=' is spelled N here
					for (int32_t k = 0; k < l_seq; k++) arena_.push_back("NACMGRSVTWYHKDBN"[(sq[k / 2] >> (4 * (1 - (k % 2)))) & 0xf]);
					r.qual_off = arena_.size(); r.qual_len = (size_t)l_seq; r.has_qual = true;
					for (int32_t k = 0; k < l_seq; k++) arena_.push_back((char)(bam_rec_[off + (size_t)k] + 33));
					off += (size_t)l_seq;
					r.tag_off = arena_.size(); r.tag_len = opt_.preserve_tags ? block_size - off : 0;
					if (r.tag_len) arena_.append(bam_rec_.data() + off, r.tag_len);
					if (pt) orig_.append(bam_rec_.data(), block_size);
					got = true;
				}
				if (!got) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
			} else if (opt_.format == 6) {             // -F k:<len>,i:<freq>: every <freq>-th <len>-mer of a FASTA file (FastaContinuousPatternSource, pat.cpp:913-1036)
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				bool emitted = false;
				while (!emitted) {
					if (fc_pos_ >= fc_line_.size()) {
						if (!src_.next(p, n)) break;
						if (n && p[0] == '>') {
							size_t k = 1;
							fc_prefix_.clear();
							while (k < n && !isspace((unsigned char)p[k])) fc_prefix_.push_back(p[k++]);
							fc_prefix_.push_back('_');
							fc_eat_ = (size_t)opt_.fc_len - 1; fc_beginning_ = true; fc_win_.clear(); fc_last_ = fc_cur_;
							fc_line_.clear(); fc_pos_ = 0;
							continue;
						}
						fc_line_.assign(p, n); fc_pos_ = 0;
						continue;
					}
					const unsigned char ch = (unsigned char)fc_line_[fc_pos_++];
					int cat = 0;
					switch (toupper(ch)) {
						case 'A': case 'C': case 'G': case 'T': cat = 1; break;
						case 'B': case 'D': case 'H': case 'K': case 'M': case 'N': case 'R': case 'S': case 'V': case 'W': case 'X': case 'Y': case '-': cat = 2; break;
						default: cat = 0;
					}
					if (cat == 0) continue;
					fc_win_.push_back(cat >= 2 ? 'N' : (char)ch);
					if (fc_win_.size() > (size_t)opt_.fc_len) fc_win_.erase(0, fc_win_.size() - (size_t)opt_.fc_len);
					if (fc_eat_ > 0) { fc_eat_--; if (!fc_beginning_) fc_cur_++; continue; }
					const std::string nm = fc_prefix_ + std::to_string(fc_cur_ - fc_last_);
					r.name_off = arena_.size(); r.name_len = nm.size(); arena_.append(nm);
					r.seq_off = arena_.size(); r.seq_len = fc_win_.size(); arena_.append(fc_win_);
					if (pt) { orig_ += nm; orig_.push_back('\t'); orig_ += fc_win_; }
					fc_eat_ = (size_t)opt_.fc_freq - 1; fc_cur_++; fc_beginning_ = false;
					emitted = true;
				}
				if (!emitted) { b.last = true; break; }
			} else {                                   // raw: one sequence per line, named by its index
				bool got;
				do { got = src_.next(p, n); } while (got && n == 0);
				if (!got) { b.last = true; break; }
				if ((rdid_ >> ushift_) - std::min<uint64_t>((rdid_ >> ushift_), opt_.skip) >= opt_.upto) { b.last = true; b.upto_hit = true; break; }
				r.name_off = arena_.size(); r.name_len = 0;
				r.seq_off = arena_.size(); r.seq_len = n; arena_.append(p, n);
				if (pt) orig_.append(p, n);
			}
			have_record:
			// a FASTA record without a single sequence character is not a read for the reference ("FASTA ended prematurely", pat.cpp:849-851):
			// it takes its number in the input and is dropped (unpaired input only: a pair loses both mates there)
			if (opt_.format == 1 && !opt_.paired && r.seq_len == 0) { orig_.resize(r.orig_off); rdid_++; continue; }
			r.orig_len = orig_.size() - r.orig_off;
			r.rdid = (rdid_ >> ushift_);
			const bool skip1 = ((rdid_++) >> ushift_) < opt_.skip;
			if (!skip1) recs_.push_back(r);
			if (have_r2) {
				have_r2 = false;
				r2.rdid = (rdid_ >> ushift_);
				if (!(((rdid_++) >> ushift_) < opt_.skip)) recs_.push_back(r2);
			}
			if (skip1) continue;
		}
		b.io_error = src_.io_error();
		t_split += tnow() - t0_;
	}
	void finish(RawBatch& rb, HostBatch& b, size_t max_read_len) {
		const double t1_ = tnow();
		const std::string& arena_ = rb.arena; const std::vector<Raw>& recs_ = rb.recs; const std::string& orig_ = rb.orig;
		const bool pt = opt_.passthrough;
		if (len_params_.empty()) { const size_t nl = std::min<size_t>(max_read_len, 4096) + 1; len_params_.resize(nl); for (size_t l = 0; l < nl; l++) len_params_[l] = compute_len_params(opt_, l); }
		b.last = rb.last; b.upto_hit = rb.upto_hit; b.bad_input = rb.bad_input;
		// ---- parallel part 1: records -> names, codes and qualities in per-chunk arenas (no per-read allocations)
		const size_t nrec = recs_.size();
		const size_t chunk = 4096, nchunks = (nrec + chunk - 1) / chunk;
		b.reads.assign(nrec, ReadRec());
		b.rp.resize(nrec);
		// (a recycled batch keeps its chunks' memory: see HostBatch::recycle)
		b.chunks.resize(nchunks);
		for (HostBatch::Chunk& c : b.chunks) { c.names.clear(); c.seq.clear(); c.qual.clear(); c.orig.clear(); c.tags.clear(); c.error.clear(); }
		std::vector<uint32_t> name_off(nrec), name_len(nrec), rlen(nrec), orig_off(pt ? nrec : 0), orig_len(pt ? nrec : 0);
		const bool tg = opt_.preserve_tags;
		std::vector<uint32_t> tag_off(tg ? nrec : 0), tag_len(tg ? nrec : 0);
		parallel_for(nchunks, threads_, [&](size_t c) {
			HostBatch::Chunk& ch = b.chunks[c];
			std::string tseq, tqual;
			const size_t e = std::min(nrec, (c + 1) * chunk);
			for (size_t i = c * chunk; i < e; i++) {
				const Raw& r = recs_[i];
				tseq.clear(); tqual.clear();
				const unsigned char* s = (const unsigned char*)arena_.data() + r.seq_off;
				// letters and '.' are bases (asc2dna: A/C/G/T either case -> 0..3, anything else -> 4), every other character is dropped
				tseq.resize(r.seq_len);
				{
					char* o = &tseq[0];
					size_t m = 0;
					for (size_t k = 0; k < r.seq_len; k++) { const unsigned char v = kSeqCode.t[s[k]]; o[m] = (char)v; m += (v != 255); }
					tseq.resize(m);
				}
				if (r.has_qual) {
					tqual.assign(arena_.data() + r.qual_off, r.qual_len);
					if (opt_.format == 7) {}          // BAM qualities are Phred values, whatever scale options say (pat.cpp:1493)
					else if (opt_.solexa_quals) {
						// charToPhred33 with solQuals (qual.h:112-123): Solexa log-odds Q_s = c - 64 -> Phred = 10 log10(1 + 10^(Q_s/10)), rounded;
						// the two scales agree from 10 up, and anything below -10 is Phred 0
						for (char& q : tqual) {
							const int sol = (int)q - 64;
							const int ph = sol >= 10 ? sol : sol < -10 ? 0 : (int)(10.0 * log10(1.0 + pow(10.0, sol / 10.0)) + 0.5);
							q = (char)(ph + 33);
						}
					} else if (opt_.phred64) for (char& q : tqual) q = (char)((int)q - 64 + 33 < 33 ? 33 : (int)q - 64 + 33);
					// tooFewQualities / tooManyQualities (pat.cpp:1736-1748): the reference aborts, and so do we
					if (tqual.size() != tseq.size() && ch.error.empty()) {
						const std::string nm = r.name_len ? std::string(arena_.data() + r.name_off, r.name_len) : std::to_string(r.rdid);
						ch.error = tqual.size() < tseq.size() ? "Read " + nm + " has more read characters than quality values."
						                                      : "Read " + nm + " has more quality values than read characters.";
					}
					if (tqual.size() > tseq.size()) tqual.resize(tseq.size());
					while (tqual.size() < tseq.size()) tqual.push_back('I');
				} else tqual.assign(tseq.size(), 'I');
				// -5/-3 hard trimming (pat.cpp:726-765)
				if (opt_.trim5 > 0 || opt_.trim3 > 0) {
					const size_t t5 = std::min<size_t>((size_t)opt_.trim5, tseq.size());
					tseq.erase(0, t5); tqual.erase(0, t5);
					const size_t t3 = std::min<size_t>((size_t)opt_.trim3, tseq.size());
					tseq.resize(tseq.size() - t3); tqual.resize(tqual.size() - t3);
				}
				// --trim-to: cut reads longer than the limit from the chosen end
				if (opt_.trim_to_len >= 0 && tseq.size() > (size_t)opt_.trim_to_len) {
					const size_t cut = tseq.size() - (size_t)opt_.trim_to_len;
					if (opt_.trim_to_side == 5) { tseq.erase(0, cut); tqual.erase(0, cut); }
					else { tseq.resize(opt_.trim_to_len); tqual.resize(opt_.trim_to_len); }
				}
				name_off[i] = (uint32_t)ch.names.size();
				if (r.name_len) ch.names.append(arena_.data() + r.name_off, r.name_len);
				else ch.names += std::to_string(r.rdid);
				name_len[i] = (uint32_t)ch.names.size() - name_off[i];
				rlen[i] = (uint32_t)tseq.size();
				ch.seq += tseq; ch.qual += tqual;
				if (pt) { orig_off[i] = (uint32_t)ch.orig.size(); ch.orig.append(orig_.data() + r.orig_off, r.orig_len); orig_len[i] = (uint32_t)r.orig_len; }
				if (tg) { tag_off[i] = (uint32_t)ch.tags.size(); ch.tags.append(arena_.data() + r.tag_off, r.tag_len); tag_len[i] = (uint32_t)r.tag_len; }
				b.reads[i].filter = r.filter;
			}
		});
		const double t2_ = tnow();
		t_parse += t2_ - t1_;
		// ---- pack the device arrays: a chunk's arenas are already the packed layout of its reads
		b.off.resize(nrec + 1);
		b.off[0] = 0;
		b.max_len = 0;
		for (size_t i = 0; i < nrec; i++) {
			const size_t L = rlen[i];
			if (L > b.max_len) b.max_len = (uint32_t)L;
			b.off[i + 1] = b.off[i] + L;
		}
		b.seq.resize(b.off[nrec]); b.qual.resize(b.off[nrec]);
		parallel_for(nchunks, threads_, [&](size_t c) {
			const HostBatch::Chunk& ch = b.chunks[c];
			const size_t i0 = c * chunk, e = std::min(nrec, (c + 1) * chunk);
			if (!ch.seq.empty()) { memcpy(&b.seq[b.off[i0]], ch.seq.data(), ch.seq.size()); memcpy(&b.qual[b.off[i0]], ch.qual.data(), ch.qual.size()); }
			for (size_t i = i0; i < e; i++) {
				ReadRec& rd = b.reads[i];
				rd.name.set(ch.names.data() + name_off[i], name_len[i]);
				if (pt) rd.orig.set(ch.orig.data() + orig_off[i], orig_len[i]);
				if (tg) rd.tags.set(ch.tags.data() + tag_off[i], tag_len[i]);
				rd.seq.set((const char*)b.seq.data() + b.off[i], rlen[i]);
				rd.qual.set((const char*)b.qual.data() + b.off[i], rlen[i]);
				b.rp[i] = rlen[i] < len_params_.size() ? compute_read_params(opt_, rd, len_params_[rlen[i]]) : compute_read_params(opt_, rd);
			}
		});
		for (size_t i = 0; i < nrec; i++) if (rlen[i] > max_read_len) { b.too_long = b.reads[i].name.str(); break; }
		if (b.bad_input.empty()) for (const HostBatch::Chunk& ch : b.chunks) if (!ch.error.empty()) { b.bad_input = ch.error; break; }
		if (b.bad_input.empty() && rb.io_error) b.bad_input = "error while reading the reads file (corrupt or truncated compressed input?)";
		t_pack += tnow() - t2_;
	}
	bool fasta_started_ = false;
	bool fastq_started_ = false;     // a FASTQ record (or a line that should have been one) has been seen
	BamStream bam_; bool bam_ok_ = false; std::string bam_err_, bam_rec_; int bam_mate_ = 0;     // -b state
	std::string fc_line_, fc_prefix_, fc_win_;     // -F state
	size_t fc_pos_ = 0, fc_eat_ = 0; uint64_t fc_cur_ = 0, fc_last_ = 0; bool fc_beginning_ = true;
	std::string pending_raw_;   // --passthrough: the text of a FASTA header read ahead (Read::readOrigBuf)
	LineSource src_;
	const Options& opt_;
	unsigned threads_;
	std::string pending_;
	bool have_pending_ = false;
	size_t cmd_pos_ = 0;
	std::string cmd_;            // -c: the comma-separated reads of this source (the -U, -1 or -2 argument)
	uint64_t rdid_ = 0;
	unsigned ushift_ = 0;        // log2(unit_)
	uint64_t unit_ = 1;          // records per read id: 2 for --interleaved mates (-s/-u and default names count pairs)
};

// --interleaved: the batch already holds mate 1 / mate 2 alternating; mark it paired and apply the pair-level interval adjustment
inline void finalize_interleaved(HostBatch& b, const Options& opt) {
	b.paired = true;
	if (b.bad_input.empty() && (b.reads.size() & 1)) b.bad_input = "odd number of records in the --interleaved input";
	for (size_t i = 0; i + 1 < b.reads.size(); i += 2) {
		if ((b.rp[i].filt & 15u) == 15u && (b.rp[i + 1].filt & 15u) == 15u) {
			for (int m = 0; m < 2; m++) {
				int iv = opt.ms_ival.f<int>((double)b.reads[i + m].seq.size());
				iv = (int)(iv * 1.2 + 0.5);
				b.rp[i + m].interval = iv < 1 ? 1 : iv;
			}
		}
	}
}

// SAM text of one batch, formatted in parallel chunks (concatenate `parts` in order), plus the batch's share of the
// alignment summary and the reads the device flagged.  `parts` keeps its capacity from batch to batch.
struct BatchTally {
	AlnSummary summ;
	PairSummary psumm;
	std::vector<size_t> flagged;
};
inline void format_batch(const HostBatch& b, const Options& opt, const RefInfo& ref, unsigned threads, std::vector<std::string>& parts, BatchTally& tally) {
	const size_t n = b.reads.size(), chunk = 2048, nchunks = (n + chunk - 1) / chunk;
	if (parts.size() < nchunks) parts.resize(nchunks);
	for (std::string& o : parts) o.clear();
	std::vector<BatchTally> tl(nchunks);
	parallel_for(nchunks, threads, [&](size_t c) {
		std::string& o = parts[c];
		if (o.capacity() < chunk * 400) o.reserve(chunk * 400);
		const size_t e = std::min(n, (c + 1) * chunk);
		if (b.paired) {
			// mates travel together: chunk boundaries are even, records 2p / 2p+1 make pair p
			std::vector<const AlnRes*> a1, a2;
			for (size_t i = c * chunk; i + 1 < e; i += 2) {
				const ReadResult& r1 = b.result(i); const ReadResult& r2 = b.result(i + 1);
				tl[c].psumm.add(r1, r2);
				if (r1.status || r2.status) tl[c].flagged.push_back(i);
				a1.clear(); a2.clear();
				const AlnRes* a = &r1.alns[0];
				for (uint32_t k = 0; k < r1.nreport; k++, a = HostBatch::next_aln(a)) a1.push_back(a);
				a = &r2.alns[0];
				for (uint32_t k = 0; k < r2.nreport; k++, a = HostBatch::next_aln(a)) a2.push_back(a);
				sam_pair_records(o, opt, ref, b.reads[i], b.reads[i + 1], r1, r2, a1.data(), a2.data());
			}
			return;
		}
		for (size_t i = c * chunk; i < e; i++) {
			const ReadResult& rr = b.result(i);
			tl[c].summ.add(rr);
			if (rr.status) tl[c].flagged.push_back(i);
			if (rr.aligned) {
				const AlnRes* a = &rr.alns[0];
				for (uint32_t k = 0; k < rr.nreport; k++, a = HostBatch::next_aln(a)) sam_record(o, opt, ref, b.reads[i], rr, a, k == 0);
			} else if (!opt.no_unal) sam_record(o, opt, ref, b.reads[i], rr, nullptr, true);
		}
	});
	for (const BatchTally& t : tl) { tally.summ.merge(t.summ); tally.psumm.merge(t.psumm); tally.flagged.insert(tally.flagged.end(), t.flagged.begin(), t.flagged.end()); }
}

} // namespace bt2g
#endif
