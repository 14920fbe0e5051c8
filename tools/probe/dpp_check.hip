// dpp_check: run on the GPU box before a long session -- checks the DPP controls the band fill relies on
// (wave_shl:1, wave_shr:1, row_shr:n, row_bcast:15/31 with zero fill) against their lane-index definitions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true); }
__global__ void k(uint32_t* out, uint32_t D) {
	const uint32_t lane = threadIdx.x;
	const uint32_t v = 100 + lane;
	out[0 * 64 + lane] = dpp0<0x130, 0xf>(v);     // wave_shl:1  -> lane+1
	out[1 * 64 + lane] = dpp0<0x138, 0xf>(v);     // wave_shr:1  -> lane-1
	out[2 * 64 + lane] = dpp0<0x111, 0xf>(v);     // row_shr:1
	out[3 * 64 + lane] = dpp0<0x118, 0xf>(v);     // row_shr:8
	out[4 * 64 + lane] = dpp0<0x142, 0xa>(v);     // row_bcast:15 rows 1,3
	out[5 * 64 + lane] = dpp0<0x143, 0xc>(v);     // row_bcast:31 rows 2,3
	// the decaying max scan
	auto subs = [](uint32_t a, uint32_t b) { return a > b ? a - b : 0u; };
	auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
	uint32_t X = (lane * 2654435761u >> 24);      // 0..255
	out[6 * 64 + lane] = X;
	X = mx(X, subs(dpp0<0x111, 0xf>(X), D));
	X = mx(X, subs(dpp0<0x112, 0xf>(X), 2 * D));
	X = mx(X, subs(dpp0<0x114, 0xf>(X), 4 * D));
	X = mx(X, subs(dpp0<0x118, 0xf>(X), 8 * D));
	X = mx(X, subs(dpp0<0x142, 0xa>(X), ((lane & 15) + 1) * D));
	X = mx(X, subs(dpp0<0x143, 0xc>(X), (lane > 31 ? lane - 31 : 0) * D));
	out[7 * 64 + lane] = X;
}
int main() {
	uint32_t* d; hipMalloc(&d, 8 * 64 * 4);
	uint32_t h[8 * 64];
	int bad = 0;
	for (uint32_t D : {1u, 3u, 6u, 40u}) {
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, D);
		hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
		for (int l = 0; l < 64; l++) {
			auto v = [](int x) { return (uint32_t)(100 + x); };
			const uint32_t e0 = l < 63 ? v(l + 1) : 0, e1 = l > 0 ? v(l - 1) : 0, e2 = (l & 15) >= 1 ? v(l - 1) : 0, e3 = (l & 15) >= 8 ? v(l - 8) : 0;
			const uint32_t e4 = ((l >> 4) & 1) ? v((l & ~15) - 1) : 0, e5 = l >= 32 ? v(31) : 0;
			if (h[l] != e0 || h[64 + l] != e1 || h[128 + l] != e2 || h[192 + l] != e3 || h[256 + l] != e4 || h[320 + l] != e5) { bad++; printf("lane %d: %u/%u %u/%u %u/%u %u/%u %u/%u %u/%u\n", l, h[l], e0, h[64 + l], e1, h[128 + l], e2, h[192 + l], e3, h[256 + l], e4, h[320 + l], e5); }
			uint32_t want = 0;
			for (int k2 = 0; k2 <= l; k2++) { const uint32_t t = h[6 * 64 + k2], dec = (uint32_t)(l - k2) * D; const uint32_t s = t > dec ? t - dec : 0; if (s > want) want = s; }
			if (h[7 * 64 + l] != want) { bad++; printf("scan D=%u lane %d: got %u want %u\n", D, l, h[7 * 64 + l], want); }
		}
	}
	printf(bad ? "dpp_check: %d MISMATCHES\n" : "dpp_check: ok\n", bad);
	return bad ? 1 : 0;
}
