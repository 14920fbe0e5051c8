// bt2g_rankidx.hip -- kernels that build the HBM layout of the index (bt2g_rankidx.hpp) from the uploaded file sections.
//   k_make_rank_blocks   one lane per 128-character block
//   k_sa_init            one lane per row: SA samples and the '$' row
//   k_sa_segments        one lane per segment head: the reference's getOffset walk, every row visited once
// Run once per bt2g_index_load; the verbatim sides and the SA sample are freed afterwards.
#include "bt2g_kernels.hpp"
#include "bt2g_rankidx.hpp"

namespace bt2g {

template <typename TOff> struct FchrArg { TOff v[5]; };

template <typename TOff>
__global__ void __launch_bounds__(256)
k_make_rank_blocks(const uint8_t* __restrict__ ebwt, uint64_t n_sides, FchrArg<TOff> fchr, TOff zoff, RankBlock* __restrict__ out, uint64_t n_blocks) {
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= n_blocks) return;
	RankBlock r;
	make_rank_block<TOff>(ebwt, n_sides, b, fchr.v, zoff, r);
	out[b] = r;
}

template <typename TOff>
__global__ void __launch_bounds__(256)
k_sa_init(DevEbwt<TOff> e, const TOff* __restrict__ offs, uint64_t* __restrict__ sa) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r > (uint64_t)e.len) return;
	sa[r] = kJoffNone;
	sa_init_row(e, offs, (TOff)r, sa);
}

template <typename TOff>
__global__ void __launch_bounds__(256)
k_sa_segments(DevEbwt<TOff> e, const TOff* __restrict__ offs, uint64_t* __restrict__ sa, uint64_t n_heads, unsigned long long* __restrict__ lost) {
	const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (h >= n_heads) return;
	uint32_t l = 0;
	sa_segment(e, offs, sa_head_row(e, h), sa, &l);
	if (l) atomicAdd(lost, (unsigned long long)l);      // rows whose walk is longer than joff_pack's step field (see sa_segment)
}

template <typename TOff>
hipError_t launch_make_rank_blocks(const uint8_t* d_ebwt, uint64_t n_sides, const TOff fchr[5], TOff zoff, RankBlock* d_out, uint64_t n_blocks, hipStream_t st) {
	FchrArg<TOff> f;
	for (int i = 0; i < 5; i++) f.v[i] = fchr[i];
	const uint64_t grid = (n_blocks + 255) / 256;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_make_rank_blocks<TOff>, dim3((uint32_t)grid), dim3(256), 0, st, d_ebwt, n_sides, f, zoff, d_out, n_blocks);
	return hipGetLastError();
}

template <typename TOff>
hipError_t launch_make_full_sa(const DevEbwt<TOff>& e, const TOff* d_offs, uint64_t* d_sa, unsigned long long* d_lost, hipStream_t st) {
	const uint64_t n_rows = (uint64_t)e.len + 1;
	uint64_t grid = (n_rows + 255) / 256;
	if (grid > 0x7fffffffull) return hipErrorInvalidValue;
	hipLaunchKernelGGL(k_sa_init<TOff>, dim3((uint32_t)grid), dim3(256), 0, st, e, d_offs, d_sa);
	const uint64_t nh = sa_n_heads(e);
	grid = (nh + 255) / 256;
	hipLaunchKernelGGL(k_sa_segments<TOff>, dim3((uint32_t)grid), dim3(256), 0, st, e, d_offs, d_sa, nh, d_lost);
	return hipGetLastError();
}

template hipError_t launch_make_rank_blocks<uint32_t>(const uint8_t*, uint64_t, const uint32_t*, uint32_t, RankBlock*, uint64_t, hipStream_t);
template hipError_t launch_make_rank_blocks<uint64_t>(const uint8_t*, uint64_t, const uint64_t*, uint64_t, RankBlock*, uint64_t, hipStream_t);
template hipError_t launch_make_full_sa<uint32_t>(const DevEbwt<uint32_t>&, const uint32_t*, uint64_t*, unsigned long long*, hipStream_t);
template hipError_t launch_make_full_sa<uint64_t>(const DevEbwt<uint64_t>&, const uint64_t*, uint64_t*, unsigned long long*, hipStream_t);

} // namespace bt2g
