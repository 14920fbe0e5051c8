// tests/hostsim/build_hostsim.cpp -- TEST-ONLY CPU twin of the index builder.
//
// The builder's logic (bowtie2_amd/csrc/bt2g_build_*.hpp) is written against a small set of primitives; the product
// binds them to rocPRIM and HIP kernels (bt2g_build.hip).  This file binds the same source to std:: algorithms so that
// the CPU test suite can check the emitted files byte for byte against the reference's bowtie2-build without a GPU.
// Never shipped, never linked into the product.
//   build_hostsim [--large-index] [-c] [-o N] [-t N] <reference_in> <bt2_index_base>
#define BT2_HD_LAMBDA
#include "../../bowtie2_amd/csrc/bt2g_build_cli.hpp"

#include <algorithm>
#include <numeric>

namespace {

struct HostBk {
	static const char* name() { return "host twin"; }
	static const std::string& error() { static const std::string none; return none; }
	static bool init(int, std::string&) { return true; }
	template <typename T> static T* alloc(uint64_t n) { return static_cast<T*>(malloc((n ? n : 1) * sizeof(T))); }
	static void release(void* p) { free(p); }
	static void upload(void* d, const void* s, uint64_t n) { memcpy(d, s, n); }
	static void download(void* d, const void* s, uint64_t n) { memcpy(d, s, n); }
	static void copy(void* d, const void* s, uint64_t n) { memmove(d, s, n); }
	template <class F> static void pfor(uint64_t n, F f) { for (uint64_t i = 0; i < n; i++) f(i); }
	template <typename K, typename V>
	static void sort_pairs(bt2g::build::DBuf<K>& k, bt2g::build::DBuf<V>& v, uint64_t n, int b0, int b1) {
		const K mask = (b1 - b0 >= 64 ? ~(K)0 : (((K)1 << (b1 - b0)) - 1)) << b0;
		std::vector<uint64_t> perm(n);
		std::iota(perm.begin(), perm.end(), 0);
		const K* kc = k.cur;
		std::stable_sort(perm.begin(), perm.end(), [&](uint64_t a, uint64_t b) { return (kc[a] & mask) < (kc[b] & mask); });
		for (uint64_t i = 0; i < n; i++) { k.alt[i] = k.cur[perm[i]]; v.alt[i] = v.cur[perm[i]]; }
		std::swap(k.cur, k.alt); std::swap(v.cur, v.alt);
	}
	template <typename T> static void inclusive_max(T* a, uint64_t n) { for (uint64_t i = 1; i < n; i++) if (a[i] < a[i - 1]) a[i] = a[i - 1]; }
	static void exclusive_sum(uint64_t* a, uint64_t n) { uint64_t acc = 0; for (uint64_t i = 0; i < n; i++) { const uint64_t v = a[i]; a[i] = acc; acc += v; } }
	template <typename T> static uint64_t select_index(const uint8_t* f, T* out, uint64_t n) { uint64_t m = 0; for (uint64_t i = 0; i < n; i++) if (f[i]) out[m++] = (T)i; return m; }
	template <typename T> static uint64_t select(const T* in, const uint8_t* f, T* out, uint64_t n) { uint64_t m = 0; for (uint64_t i = 0; i < n; i++) if (f[i]) out[m++] = in[i]; return m; }
	static void fill0(uint64_t* p, uint64_t n) { memset(p, 0, n * 8); }
	static void atomic_add(uint64_t* p, uint64_t v) { *p += v; }
};

} // namespace

int main(int argc, const char** argv) {
	bool force64 = false;      // --idx64: run the 64-bit-position code path on a small text (test hook)
	std::vector<const char*> av;
	for (int i = 0; i < argc; i++) { if (std::string(argv[i]) == "--idx64") force64 = true; else av.push_back(argv[i]); }
	if (force64) setenv("BT2G_BUILD_FORCE_IDX64", "1", 1);
	return bt2g::build::build_main<HostBk>((int)av.size(), av.data(), false);
}
