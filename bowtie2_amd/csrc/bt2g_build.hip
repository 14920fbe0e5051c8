// bt2g_build.hip -- device backend of the index builder (gfx950) + its C ABI (include/bt2g.h: bt2g_index_build*).
//
// The algorithm lives in bt2g_build_core.hpp; here its primitives are bound to the GPU: grid-stride kernels for the
// element-wise steps and rocPRIM's device-wide radix sort / scan / select for the rest.  All arrays (packed text, keys,
// suffix array, inverse suffix array: ~30 bytes per base at the peak) sit in HBM; a 3.1 Gbp genome needs ~95 GB of the
// 288 GB.  No host fallback: without a gfx950 device the entry points return BT2G_ERR_NO_DEVICE.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#define BT2_HD_LAMBDA __host__ __device__
#include "bt2g_build_cli.hpp"
#include "../../include/bt2g.h"

namespace bt2g { namespace build {

template <class F>
__global__ void __launch_bounds__(256) k_pfor(uint64_t n, F f) {
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) f(i);
}

struct DevBk {
	static const char* name() { return "MI355X"; }
	static bool init(int device, std::string& err) {
		int nd = 0;
		if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device < 0 || device >= nd) { err = "Error: no usable HIP device (this builder has no CPU path)"; return false; }
		if (hipSetDevice(device) != hipSuccess) { err = "Error: hipSetDevice failed"; return false; }
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { err = "Error: device is not gfx950 (MI355X)"; return false; }
		return true;
	}
	template <typename T> static T* alloc(uint64_t n) { void* p = nullptr; if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return nullptr; } return static_cast<T*>(p); }
	static void release(void* p) { if (p) (void)hipFree(p); }
	static void upload(void* d, const void* s, uint64_t n) { if (n) check(hipMemcpy(d, s, n, hipMemcpyHostToDevice), "hipMemcpy H2D"); }
	static void download(void* d, const void* s, uint64_t n) { if (n) check(hipMemcpy(d, s, n, hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
	static void copy(void* d, const void* s, uint64_t n) { if (n) check(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice), "hipMemcpy D2D"); }
	template <class F> static void pfor(uint64_t n, F f) {
		if (n == 0) return;
		uint64_t blocks = (n + 255) / 256;
		if (blocks > 256ull * 64) blocks = 256ull * 64;          // 64 workgroups per CU, grid-stride beyond that
		hipLaunchKernelGGL(k_pfor<F>, dim3((unsigned)blocks), dim3(256), 0, 0, n, f);
		check(hipGetLastError(), "kernel launch");
	}
	template <typename K, typename V>
	static void sort_pairs(DBuf<K>& k, DBuf<V>& v, uint64_t n, int b0, int b1) {
		if (n == 0) return;
		rocprim::double_buffer<K> kb(k.cur, k.alt);
		rocprim::double_buffer<V> vb(v.cur, v.alt);
		size_t bytes = 0;
		check(rocprim::radix_sort_pairs(nullptr, bytes, kb, vb, (size_t)n, (unsigned)b0, (unsigned)b1, (hipStream_t)0), "radix_sort_pairs(size)");
		void* tmp = scratch(bytes);
		check(rocprim::radix_sort_pairs(tmp, bytes, kb, vb, (size_t)n, (unsigned)b0, (unsigned)b1, (hipStream_t)0), "radix_sort_pairs");
		k.cur = kb.current(); k.alt = kb.alternate(); v.cur = vb.current(); v.alt = vb.alternate();
	}
	template <typename T> static void inclusive_max(T* a, uint64_t n) {
		if (n == 0) return;
		size_t bytes = 0;
		check(rocprim::inclusive_scan(nullptr, bytes, a, a, (size_t)n, rocprim::maximum<T>(), (hipStream_t)0), "inclusive_scan(size)");
		void* tmp = scratch(bytes);
		check(rocprim::inclusive_scan(tmp, bytes, a, a, (size_t)n, rocprim::maximum<T>(), (hipStream_t)0), "inclusive_scan");
	}
	static void exclusive_sum(uint64_t* a, uint64_t n) {
		if (n == 0) return;
		size_t bytes = 0;
		check(rocprim::exclusive_scan(nullptr, bytes, a, a, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t)0), "exclusive_scan(size)");
		void* tmp = scratch(bytes);
		check(rocprim::exclusive_scan(tmp, bytes, a, a, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), (hipStream_t)0), "exclusive_scan");
	}
	template <typename T> static uint64_t select_index(const uint8_t* f, T* out, uint64_t n) {
		return select_impl(rocprim::counting_iterator<T>((T)0), f, out, n);
	}
	template <typename T> static uint64_t select(const T* in, const uint8_t* f, T* out, uint64_t n) { return select_impl(in, f, out, n); }
	static void fill0(uint64_t* p, uint64_t n) { if (n) check(hipMemset(p, 0, n * 8), "hipMemset"); }
	static __host__ __device__ __forceinline__ void atomic_add(uint64_t* p, uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
		atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
#else
		*p += v;
#endif
	}
	static std::string& last_error() { static std::string e; return e; }
	static const std::string& error() { return last_error(); }      // first device / allocation error since the build started ("" = none)
private:
	template <class In, typename T> static uint64_t select_impl(In in, const uint8_t* f, T* out, uint64_t n) {
		if (n == 0) return 0;
		size_t bytes = 0;
		size_t* d_cnt = reinterpret_cast<size_t*>(scratch_small());
		check(rocprim::select(nullptr, bytes, in, f, out, d_cnt, (size_t)n, (hipStream_t)0), "select(size)");
		void* tmp = scratch(bytes);
		check(rocprim::select(tmp, bytes, in, f, out, d_cnt, (size_t)n, (hipStream_t)0), "select");
		size_t cnt = 0;
		check(hipMemcpy(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost), "hipMemcpy(select count)");
		return (uint64_t)cnt;
	}
	static void check(hipError_t e, const char* what) { if (e != hipSuccess && last_error().empty()) last_error() = std::string(what) + ": " + hipGetErrorString(e); }
	// temporary storage of the rocPRIM calls: one buffer, grown on demand, kept for the life of the process
	static void* scratch(size_t bytes) {
		static void* p = nullptr; static size_t cap = 0;
		if (bytes > cap) {
			if (p) (void)hipFree(p);
			p = nullptr; cap = 0;
			if (hipMalloc(&p, bytes + 256) == hipSuccess) cap = bytes + 256;
			else { p = nullptr; if (last_error().empty()) last_error() = "out of device memory (rocPRIM scratch)"; }
		}
		// (with a null pointer rocPRIM would read the call as a size query and return success without doing any work: the error latched
		// above makes build_index_files stop before anything derived from that non-result reaches a file)
		return p;
	}
	static void* scratch_small() { static void* p = nullptr; if (!p) (void)hipMalloc(&p, 256); return p; }
};

} } // namespace bt2g::build

using namespace bt2g::build;

extern "C" {

void bt2g_build_params_default(bt2g_build_params* p) {
	if (!p) return;
	p->large_index = 0; p->off_rate = 4; p->ftab_chars = 10; p->write_ref = 1; p->device = 0;
}

static int run_build(RefInput& in, const char* out_base, const bt2g_build_params* bp, double t_parse, bt2g_build_stats* stats) {
	Params P;
	P.off_size = bp->large_index ? 8 : 4; P.line_rate = bp->large_index ? 7 : 6; P.off_rate = bp->off_rate; P.ftab_chars = bp->ftab_chars; P.write_ref = bp->write_ref != 0;
	if (P.ftab_chars < 1 || P.ftab_chars > 15 || P.off_rate < 0 || P.off_rate > 30) return BT2G_ERR_ARG;
	std::string err;
	if (!DevBk::init(bp->device, err)) return BT2G_ERR_NO_DEVICE;
	DevBk::last_error().clear();
	BuildStats st;
	st.t_parse = t_parse;
	const bool ok = build_index_files<DevBk>(in, out_base, P, st, err, &wall_now);
	if (stats) {
		stats->len = st.len; stats->n_pat = st.n_pat; stats->n_frag = st.n_frag; stats->rounds_fw = st.rounds_fw; stats->rounds_bw = st.rounds_bw;
		stats->tied_fw = st.tied_fw; stats->tied_bw = st.tied_bw; stats->t_parse = st.t_parse; stats->t_fw = st.t_fw; stats->t_bw = st.t_bw; stats->t_write = st.t_write;
	}
	if (!ok) { fprintf(stderr, "%s\n", err.c_str()); return err.find("memory") != std::string::npos ? BT2G_ERR_NOMEM : BT2G_ERR_IO; }
	if (!DevBk::last_error().empty()) { fprintf(stderr, "bt2g_index_build: %s\n", DevBk::last_error().c_str()); return BT2G_ERR_HIP; }
	return 0;
}

int bt2g_index_build(const char* const* fasta_paths, uint32_t n_paths, const char* out_base, const bt2g_build_params* bp, bt2g_build_stats* stats) {
	if (!fasta_paths || !n_paths || !out_base || !bp) return BT2G_ERR_ARG;
	RefInput in;
	std::string err;
	uint64_t seqs = 0;
	const double t0 = wall_now();
	for (uint32_t i = 0; i < n_paths; i++) {
		GzSource src(fasta_paths[i]);
		if (!src.ok()) return BT2G_ERR_IO;
		if (src.at_end()) { if (src.io_error()) return BT2G_ERR_IO; continue; }
		if (!scan_fasta(src, in, seqs, err)) return BT2G_ERR_FORMAT;
		if (src.io_error()) { fprintf(stderr, "bt2g_index_build: reading %s failed (corrupt or truncated compressed file?)\n", fasta_paths[i]); return BT2G_ERR_IO; }
	}
	return run_build(in, out_base, bp, wall_now() - t0, stats);
}

int bt2g_index_build_mem(const char* const* names, const char* const* seqs, const uint64_t* lens, uint32_t n_seqs, const char* out_base,
                         const bt2g_build_params* bp, bt2g_build_stats* stats) {
	if (!seqs || !lens || !n_seqs || !out_base || !bp) return BT2G_ERR_ARG;
	RefInput in;
	std::string err;
	uint64_t nseq = 0;
	const double t0 = wall_now();
	uint64_t tot = 0;
	for (uint32_t i = 0; i < n_seqs; i++) tot += lens[i];
	in.joined.reserve(tot);
	// the sequences are scanned as one FASTA stream: ">name\n" + characters + "\n" per entry
	class PiecesSource : public ByteSource {
	public:
		std::vector<std::pair<const char*, size_t>> pieces; size_t pi = 0, po = 0;
	protected:
		size_t read_some(uint8_t* dst, size_t cap) override {
			size_t got = 0;
			while (got < cap && pi < pieces.size()) {
				const size_t k = std::min(cap - got, pieces[pi].second - po);
				memcpy(dst + got, pieces[pi].first + po, k);
				got += k; po += k;
				if (po == pieces[pi].second) { pi++; po = 0; }
			}
			return got;
		}
	} src;
	std::vector<std::string> hdr(n_seqs);
	for (uint32_t i = 0; i < n_seqs; i++) {
		hdr[i] = ">" + (names && names[i] ? std::string(names[i]) : std::to_string(i)) + "\n";
		src.pieces.push_back({hdr[i].data(), hdr[i].size()});
		if (lens[i]) src.pieces.push_back({seqs[i], (size_t)lens[i]});
		src.pieces.push_back({"\n", 1});
	}
	if (!scan_fasta(src, in, nseq, err)) return BT2G_ERR_FORMAT;
	return run_build(in, out_base, bp, wall_now() - t0, stats);
}

} // extern "C"

// entry of the bowtie2-build-{s,l} executables (bt2g_build_main.cpp)
int bt2g_build_cli_main(int argc, const char** argv, int large_default) {
	const int rc = build_main<DevBk>(argc, argv, large_default != 0);
	if (rc == 0 && !DevBk::last_error().empty()) { fprintf(stderr, "bowtie2-build: %s\n", DevBk::last_error().c_str()); return 1; }
	return rc;
}

// The reference's library-style entry point of the builder, same name and signature (bt2_build.cpp:556-560).  The index width
// follows the program name in argv[0] ("...build-l" -> .bt2l), as for the executables; --large-index forces .bt2l.
extern "C" int bowtie_build(int argc, const char** argv) {
	const std::string me = argc > 0 && argv[0] ? argv[0] : "";
	const size_t sl = me.find_last_of('/');
	const std::string base = sl == std::string::npos ? me : me.substr(sl + 1);
	return bt2g_build_cli_main(argc, argv, base.find("build-l") != std::string::npos ? 1 : 0);
}
