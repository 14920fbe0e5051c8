// lazy_index_check.cpp -- test-only: load_index(base, ..., lazy = true) must describe exactly the bytes load_index(base) reads
// (tests/test_index_lazy_load.py).  Prints "ok <sections> <bytes>" or the first difference.
#include "../../bowtie2_amd/csrc/bt2g_index.hpp"
#include <cstdio>
#include <cstring>

using namespace bt2g;

static bool same(const FileSpan& sp, const std::vector<uint8_t>& v, uint64_t slack, const char* what, uint64_t& total) {
	if (sp.path.empty() || sp.nbytes + slack != v.size()) { printf("%s: span %llu bytes, vector %zu\n", what, (unsigned long long)sp.nbytes, v.size()); return false; }
	FILE* f = fopen(sp.path.c_str(), "rb");
	if (!f) { printf("%s: cannot open %s\n", what, sp.path.c_str()); return false; }
	std::vector<uint8_t> b(sp.nbytes);
	const bool ok = fseeko(f, (off_t)sp.off, SEEK_SET) == 0 && (sp.nbytes == 0 || fread(b.data(), 1, sp.nbytes, f) == sp.nbytes);
	fclose(f);
	if (!ok || memcmp(b.data(), v.data(), sp.nbytes) != 0) { printf("%s: bytes differ\n", what); return false; }
	total += sp.nbytes;
	return true;
}

int main(int argc, char** argv) {
	if (argc < 2) return 2;
	HostIndex a, b;
	std::string err;
	if (load_index(argv[1], b, err, true) != 0) { printf("lazy: %s\n", err.c_str()); return 1; }
	if (load_index(argv[1], a, err) != 0) { printf("eager: %s\n", err.c_str()); return 1; }
	if (!b.fw.ebwt.empty() || !b.fw.offs.empty() || !b.bw.ebwt.empty() || !b.ref.buf.empty()) { printf("lazy load read a large section\n"); return 1; }
	uint64_t total = 0;
	if (!same(b.fw.ebwt_span, a.fw.ebwt, 0, "fw.ebwt", total) || !same(b.fw.offs_span, a.fw.offs, 0, "fw.offs", total) ||
	    !same(b.bw.ebwt_span, a.bw.ebwt, 0, "bw.ebwt", total) || !same(b.ref.buf_span, a.ref.buf, 16, "ref.buf", total)) return 1;
	// everything else is read either way
	if (a.fw.ftab != b.fw.ftab || a.fw.eftab != b.fw.eftab || a.fw.plen != b.fw.plen || a.fw.rstarts != b.fw.rstarts || a.bw.ftab != b.bw.ftab ||
	    a.fw.zoff != b.fw.zoff || a.bw.zoff != b.bw.zoff || a.fw.refnames != b.fw.refnames || a.ref.rec_len != b.ref.rec_len || a.fw.ebwt_tot_len != b.fw.ebwt_tot_len ||
	    memcmp(a.fw.fchr, b.fw.fchr, sizeof a.fw.fchr) != 0 || memcmp(a.bw.fchr, b.bw.fchr, sizeof a.bw.fchr) != 0) { printf("small sections differ\n"); return 1; }
	printf("ok 4 %llu\n", (unsigned long long)total);
	return 0;
}
