// bt2g_index.hpp -- host-side reader for bowtie2 .bt2/.bt2l indexes (own implementation).
//
// Parses the on-disk format documented in SURVEY.md Appendix B: what the reference reads in
// Ebwt::readIntoMemory (bt2_io.cpp:39-633) and BitPairReference::BitPairReference
// (reference.cpp:30-264).  Arrays are kept in their on-disk width (4- or 8-byte offsets)
// so they can be uploaded to HBM verbatim.
#ifndef BT2G_INDEX_HPP_
#define BT2G_INDEX_HPP_

#include <cstdint>
#include <string>
#include <vector>

namespace bt2g {

// A byte range of an index file its user reads itself.  load_index(..., lazy = true) leaves the three large sections -- the BWT sides, the
// suffix-array sample, the 2-bit reference -- in their files and says where they are: bt2g_index_load streams them to the device through
// pinned buffers instead of through a second copy in pageable host memory.
struct FileSpan { std::string path; uint64_t off = 0, nbytes = 0; };

struct HostEbwt {
	uint64_t len = 0;
	int32_t  line_rate = 0, off_rate = 0, ftab_chars = 0, flags = 0;
	uint32_t side_sz = 0, side_bwt_sz = 0, side_bwt_len = 0;
	uint64_t num_sides = 0, ebwt_tot_len = 0, ftab_len = 0, eftab_len = 0, offs_len = 0;
	uint64_t n_pat = 0, n_frag = 0, zoff = 0;
	uint64_t fchr[5] = {0, 0, 0, 0, 0};
	std::vector<uint8_t> plen, rstarts, ebwt, ftab, eftab, offs; // raw little-endian arrays, on-disk width
	std::vector<std::string> refnames;
	FileSpan ebwt_span, offs_span;       // lazy load: where `ebwt` and `offs` are (the vectors stay empty)
};

struct HostRef {
	uint64_t nrecs = 0, nrefs = 0, buf_sz = 0;
	// per record (widened to u64): start position within its reference (incl. Ns),
	// start position within the 2-bit buffer, length
	std::vector<uint64_t> rec_refpos, rec_bufpos, rec_len;
	std::vector<uint64_t> ref_rec_offs; // [nrefs+1]
	std::vector<uint64_t> ref_lens;     // [nrefs]
	std::vector<uint8_t>  buf;          // 2-bit packed (+ 16 bytes of slack)
	FileSpan buf_span;                  // lazy load: where `buf` is (without the slack)
};

struct HostIndex {
	int off_size = 0;   // 4 or 8
	HostEbwt fw, bw;
	HostRef ref;
	uint64_t plen_at(uint64_t i) const;
};

// Returns 0 or a negative bt2g_status; err receives a message.
int load_index(const std::string& base, HostIndex& out, std::string& err, bool lazy = false);

} // namespace bt2g
#endif
